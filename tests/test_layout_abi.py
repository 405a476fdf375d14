"""CPU: host logic — the name -> slot layout compiler, the epoch table, and that the C-ABI library
loads and exports every symbol include/rvll.h declares (no compute calls: there is no GPU here)."""
import ctypes as C
import re
from pathlib import Path

import pytest

from evidence_amd import _abi
from evidence_amd.data import EpochTable
from evidence_amd.layout import compile_layout

REPO = Path(__file__).resolve().parents[1]
PEG = ["hamilton_jitter", "hamilton_offset", "planet1_ecc", "planet1_k1", "planet1_ma0", "planet1_omega",
       "planet1_period"]


def test_theta_order_is_sorted_names_and_51peg_shape():
    L = compile_layout(reversed(PEG), {"planet1_epoch": 51050}, ["hamilton"])
    assert L.parnames == sorted(PEG) and L.ndim == 7 and L.nplanets == 1      # tests/test_config.py:21-37: 7 free
    p = L.planets[0]
    assert (p.k_kind, p.p_kind, p.ecc_kind, p.anom_kind) == (_abi.K_K1, _abi.P_PERIOD, _abi.ECC_DIRECT, _abi.ANOM_MA0)
    assert p.k.index == L.parnames.index("planet1_k1") and p.epoch.value == 51050.0 and not p.epoch.is_free
    assert L.has_jitter and not L.has_drift and L.tref_from_data


def test_parametrisation_priority_follows_modelk():
    base = {"planet1_epoch": 0.0, "planet1_logk1": 1.0}
    L = compile_layout(["planet1_k1", "planet1_logperiod", "planet1_secos", "planet1_sesin",
                        "planet1_ecos", "planet1_esin", "planet1_ml0", "planet1_ma0", "i_offset"], base, ["i"])
    p = L.planets[0]                      # k1 before logk1; secos before ecos; ml0 before ma0 (rvmodel:412-454)
    assert (p.k_kind, p.p_kind, p.ecc_kind, p.anom_kind) == (_abi.K_K1, _abi.P_LOGPERIOD, _abi.ECC_SECOS_SESIN, _abi.ANOM_ML0)
    assert L.nplanets == 1 and not L.has_jitter
    # planets are counted over FREE names containing 'k1' (rvmodel:122-124): a free k1 AND a free logk1 of the same
    # planet count as two planets, and the reference then fails looking for planet2's parameters - so do we
    with pytest.raises(KeyError):
        compile_layout(["planet1_k1", "planet1_logk1", "planet1_period", "planet1_ecc", "planet1_omega",
                        "planet1_ma0", "i_offset"], {"planet1_epoch": 0.0}, ["i"])


def test_missing_parameters_raise_keyerror_like_the_reference():
    with pytest.raises(KeyError):
        compile_layout(["planet1_k1", "planet1_period", "i_offset"], {"planet1_epoch": 0.0}, ["i"])
    with pytest.raises(KeyError):
        compile_layout(["planet1_k1", "planet1_period", "planet1_ecc", "planet1_omega", "planet1_ma0"], {}, ["i"])


def test_fixed_overrides_free_and_flags_come_from_free_names_only():
    L = compile_layout(["i_offset", "drift_lin"], {"i_offset": 2.5, "i_jitter": 3.0, "drift_quad": 0.1}, ["i"])
    assert not L.insts[0].offset.is_free and L.insts[0].offset.value == 2.5           # dict.update, rvmodel:178
    assert not L.has_jitter                                                           # fixed jitter is ignored
    assert L.has_drift and L.drift[0].is_free and L.drift[1].value == 0.1 and L.drift[2].value == 0.0
    L2 = compile_layout(["i_offset"], {"drift_lin": 1.0}, ["i"])
    assert not L2.has_drift                                                           # no FREE name contains 'drift'


def test_layout_to_c_roundtrip():
    L = compile_layout(PEG + ["drift_lin", "drift_tref"], {"planet1_epoch": 51050}, ["hamilton"])
    c, keep = L.to_c()
    assert c.struct_size == C.sizeof(_abi.Layout) and c.ndim == 9 and c.nplanets == 1 and c.has_drift == 1
    assert c.tref_from_data == 0 and c.tref.idx == L.parnames.index("drift_tref")
    assert c.planets[0].epoch.idx == -1 and c.planets[0].epoch.val == 51050.0
    assert c.tol == 1e-4 and c.itmax == 10000                                         # rvmodel:466,491


def test_epoch_table_concatenates_by_instrument_not_time():
    dd = {"b": {"data": {"rjd": [5.0, 1.0], "vrad": [1.0, 2.0], "svrad": [0.1, 0.2]}},
          "a": {"data": {"rjd": [3.0], "vrad": [3.0], "svrad": [0.3]}}}
    t = EpochTable.from_datadict(dd)
    assert t.insts == ["b", "a"] and list(t.time) == [5.0, 1.0, 3.0] and list(t.inst_id) == [0, 0, 1]
    dd2 = {"x": {"data": {"jdb": [1.0], "vrad": [0.0], "svrad": [1.0]}}}
    assert EpochTable.from_datadict(dd2).time[0] == 1.0                               # rvmodel:141-144 fallback
    with pytest.raises(KeyError):
        EpochTable.from_datadict({"x": {"data": {"vrad": [0.0], "svrad": [1.0]}}})


def test_library_loads_and_exports_every_declared_symbol():
    lib = _abi.load()
    header = (REPO / "include" / "rvll.h").read_text()
    declared = set(re.findall(r"\b(rvll_[a-z_0-9]+)\s*\(", header))
    declared -= {"rvll_handle"}
    assert declared == set(_abi.PROTOTYPES), declared ^ set(_abi.PROTOTYPES)
    for name in declared:
        assert hasattr(lib, name)
    major, minor = C.c_int32(), C.c_int32()
    assert lib.rvll_version(C.byref(major), C.byref(minor)) == 0 and (major.value, minor.value) == _abi.ABI_VERSION


def test_struct_sizes_match_the_header_layout():
    # rvll_slot 16, rvll_planet 4*4 + 6*16, rvll_inst 32; the library re-checks rvll_layout.struct_size at create
    assert C.sizeof(_abi.Slot) == 16 and C.sizeof(_abi.Planet) == 112 and C.sizeof(_abi.Inst) == 32
    assert C.sizeof(_abi.Prior) == 8 + 8 * _abi.PRIOR_NARGS + 16 + 8


def test_product_has_no_cpu_path_and_says_so(monkeypatch, tmp_path):
    """Without a HIP device rvll_create must fail loudly (RVLL_E_NODEVICE), never fall back."""
    from evidence_amd import GpuRVModel, RvllError
    n = C.c_int32(-1)
    rc = _abi.load().rvll_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    table = EpochTable.from_arrays(["i"], [1.0, 2.0], [0.0, 0.0], [1.0, 1.0], [0, 0])
    with pytest.raises(RvllError) as err:
        GpuRVModel({}, table, ["i_offset"])
    assert err.value.code == _abi.E_NODEVICE and "no CPU path" in str(err.value)
    monkeypatch.setattr(_abi, "_lib", None)
    monkeypatch.setenv("RVLL_LIBRARY", str(tmp_path / "missing.so"))
    with pytest.raises(_abi.RvllLibraryError):
        _abi.load()


def test_oracle_is_not_imported_by_the_product():
    for path in (REPO / "evidence_amd").rglob("*.py"):
        text = path.read_text()
        assert "import oracle" not in text and "from oracle" not in text, path
    for path in (REPO / "evidence_amd" / "csrc").iterdir():
        if path.suffix in (".hip", ".h", ".cpp"):
            assert "oracle" not in path.read_text().lower(), path


def test_the_library_asks_for_its_hardware_queues_itself_unless_the_caller_has_chosen():
    """the streamed host batches' fast path must not hang on an environment variable the caller has to know
    about.  librvll's load-time constructor (csrc/rvll_comm.hip) sets GPU_MAX_HW_QUEUES=8 in the process's C environment when
    it is unset, and leaves a caller's choice alone.  (Fresh processes: the variable is read by the HIP runtime when it starts.)"""
    import subprocess
    import sys
    from evidence_amd import _abi
    code = ("import ctypes, os, sys\n"
            "os.environ.pop('GPU_MAX_HW_QUEUES', None)\n"
            "if sys.argv[2] != '-': os.environ['GPU_MAX_HW_QUEUES'] = sys.argv[2]\n"
            "ctypes.CDLL(sys.argv[1])\n"
            "libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p\n"
            "print((libc.getenv(b'GPU_MAX_HW_QUEUES') or b'unset').decode())\n")
    for chosen, want in (("-", "8"), ("2", "2")):
        out = subprocess.run([sys.executable, "-c", code, str(_abi.LIB_PATH), chosen], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert out.stdout.strip() == want, (chosen, out.stdout, out.stderr)
