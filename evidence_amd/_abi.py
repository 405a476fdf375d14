"""ctypes mirror of include/rvll.h and the loader of the in-tree librvll.so.

The product has no CPU path: if the HIP library is missing, `load()` raises
`RvllLibraryError` — it never falls back to anything else.
"""
import ctypes as C
import os
from pathlib import Path

# ---- constants (include/rvll.h) -------------------------------------------------
OK = 0
E_INVALID, E_NODEVICE, E_HIP, E_NOMEM, E_NOPRIORS, E_RCCL, E_UNSUPPORTED = -1, -2, -3, -4, -5, -6, -7
ERROR_NAMES = {
    E_INVALID: "RVLL_E_INVALID", E_NODEVICE: "RVLL_E_NODEVICE", E_HIP: "RVLL_E_HIP",
    E_NOMEM: "RVLL_E_NOMEM", E_NOPRIORS: "RVLL_E_NOPRIORS", E_RCCL: "RVLL_E_RCCL",
    E_UNSUPPORTED: "RVLL_E_UNSUPPORTED",
}

FLAG_INVALID_ORBIT = 1
FLAG_NONCONVERGED = 2
FLAG_WANDERED = 4                  # a Kepler solve took > 8 Newton steps: log-L there is conditioned to ~1e-9 (rvll.h)
ABI_VERSION = (0, 2)               # RVLL_VERSION_MAJOR / MINOR these bindings were written against

K_K1, K_LOGK1 = 0, 1
P_PERIOD, P_LOGPERIOD = 0, 1
ECC_DIRECT, ECC_SECOS_SESIN, ECC_ECOS_ESIN = 0, 1, 2
ANOM_MA0, ANOM_ML0 = 0, 1
PREC_FP64, PREC_MIXED, PREC_FP32 = 0, 1, 2
PRECISIONS = {"fp64": PREC_FP64, "mixed": PREC_MIXED, "fp32": PREC_FP32}

(PRIOR_UNIFORM, PRIOR_JEFFREYS, PRIOR_MODJEFFREYS, PRIOR_UNIFORMFREQUENCY, PRIOR_NORMAL,
 PRIOR_LOGNORMAL, PRIOR_TRUNCRAYLEIGH, PRIOR_TABLE, PRIOR_BETA, PRIOR_GAMMA, PRIOR_ALPHA,
 PRIOR_SORTED_UNIFORM, PRIOR_SORTED_LOGUNIFORM) = range(13)
PRIOR_NARGS = 6
COMM_ID_BYTES = 128


# ---- structs ---------------------------------------------------------------------
class Slot(C.Structure):
    _fields_ = [("idx", C.c_int32), ("reserved", C.c_int32), ("val", C.c_double)]

    @classmethod
    def free(cls, idx):
        return cls(int(idx), 0, 0.0)

    @classmethod
    def fixed(cls, val):
        return cls(-1, 0, float(val))


class Planet(C.Structure):
    _fields_ = [("k_kind", C.c_int32), ("p_kind", C.c_int32), ("ecc_kind", C.c_int32),
                ("anom_kind", C.c_int32),
                ("k", Slot), ("p", Slot), ("e1", Slot), ("e2", Slot), ("anom", Slot), ("epoch", Slot)]


class Inst(C.Structure):
    _fields_ = [("offset", Slot), ("jitter", Slot)]


class Layout(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("ndim", C.c_int32), ("nplanets", C.c_int32),
                ("ninst", C.c_int32), ("has_jitter", C.c_int32), ("has_drift", C.c_int32),
                ("tref_from_data", C.c_int32), ("nlinpar", C.c_int32),
                ("drift", Slot * 4), ("tref", Slot),
                ("planets", C.POINTER(Planet)), ("insts", C.POINTER(Inst)), ("linpar", C.POINTER(Slot)),
                ("tol", C.c_double), ("itmax", C.c_int32), ("precision", C.c_int32)]


class Prior(C.Structure):
    _fields_ = [("kind", C.c_int32), ("group", C.c_int32), ("args", C.c_double * PRIOR_NARGS),
                ("table_cdf", C.POINTER(C.c_double)), ("table_x", C.POINTER(C.c_double)),
                ("table_n", C.c_int32), ("table_post", C.c_int32)]


class Timing(C.Structure):
    _fields_ = [("kernel_ms_mean", C.c_double), ("kernel_ms_min", C.c_double),
                ("kernel_ms_median", C.c_double), ("total_ms", C.c_double),
                ("evals", C.c_int64), ("launches", C.c_int32), ("points_per_block", C.c_int32),
                ("blocks", C.c_int32), ("threads", C.c_int32)]


class FipTiming(C.Structure):
    _fields_ = [("index_ms", C.c_double), ("accumulate_ms", C.c_double), ("rows", C.c_int64),
                ("repeats", C.c_int32), ("reserved", C.c_int32)]


FIP_MAX_PLANETS = 8
Handle = C.c_void_p
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)

# name -> (restype, argtypes); every symbol include/rvll.h declares
PROTOTYPES = {
    "rvll_create": (C.c_int, [C.POINTER(Layout), _dp, _dp, _dp, _ip, C.c_int32, _dp, C.c_int32,
                              C.POINTER(Handle)]),
    "rvll_destroy": (C.c_int, [Handle]),
    "rvll_set_priors": (C.c_int, [Handle, C.POINTER(Prior), C.c_int32]),
    "rvll_prior_table_info": (C.c_int, [Handle, C.c_int32, _dp, _ip]),
    "rvll_slice_walk": (C.c_int, [Handle, _dp, _dp, _dp, C.c_int64, C.c_double, _dp, _ip, C.c_int32, C.c_int32,
                                  C.c_uint64, C.c_int64, C.POINTER(C.c_int64)]),
    "rvll_live_init": (C.c_int, [Handle, _dp, C.c_int64, _dp]),
    "rvll_live_step": (C.c_int, [Handle, _ip, C.c_int64, _ip, C.c_double, _dp, _ip, C.c_int32, C.c_int32, C.c_uint64,
                                 C.c_int64, C.POINTER(C.c_int64), _dp, _dp]),
    "rvll_live_sort": (C.c_int, [Handle, C.c_int64, _dp, _dp, _dp]),
    "rvll_live_get": (C.c_int, [Handle, _dp, _dp, _dp]),
    "rvll_live_dead": (C.c_int, [Handle, C.POINTER(C.c_int64), _dp, _dp]),
    "rvll_scalar_server": (C.c_int, [Handle, C.c_int32]),
    "rvll_loglike_batch": (C.c_int, [Handle, _dp, C.c_int64, _dp, _ip]),
    "rvll_prior_batch": (C.c_int, [Handle, _dp, C.c_int64, _dp]),
    "rvll_prior_loglike_batch": (C.c_int, [Handle, _dp, C.c_int64, _dp, _dp, _ip]),
    "rvll_dev_reserve": (C.c_int, [Handle, C.c_int64]),
    "rvll_dev_upload_theta": (C.c_int, [Handle, _dp, C.c_int64]),
    "rvll_dev_upload_cube": (C.c_int, [Handle, _dp, C.c_int64]),
    "rvll_dev_fill_cube": (C.c_int, [Handle, C.c_int64, C.c_uint64]),
    "rvll_dev_prior": (C.c_int, [Handle, C.c_int64]),
    "rvll_dev_loglike": (C.c_int, [Handle, C.c_int64]),
    "rvll_dev_prior_loglike": (C.c_int, [Handle, C.c_int64]),
    "rvll_dev_download": (C.c_int, [Handle, C.c_int64, _dp, _dp, _ip]),
    "rvll_dev_sync": (C.c_int, [Handle]),
    "rvll_dev_flip_lane": (C.c_int, [Handle]),
    "rvll_dev_mark": (C.c_int, [Handle, C.c_int32]),
    "rvll_dev_mark_elapsed": (C.c_int, [Handle, _dp]),
    "rvll_dev_time_loglike": (C.c_int, [Handle, C.c_int64, C.c_int32, C.c_int32, C.POINTER(Timing)]),
    "rvll_set_points_per_block": (C.c_int, [Handle, C.c_int32]),
    "rvll_set_kernel_form": (C.c_int, [Handle, C.c_int32]),
    "rvll_set_slim_table_range": (C.c_int, [Handle, C.c_double]),
    "rvll_set_walk_speculation": (C.c_int, [Handle, C.c_int32]),
    "rvll_slice_walk_evaluated": (C.c_int, [Handle, C.POINTER(C.c_int64)]),
    "rvll_slice_walk_phases": (C.c_int, [Handle, C.POINTER(C.c_uint64)]),
    "rvll_slice_walk_rounds": (C.c_int, [Handle, C.POINTER(C.c_int32)]),
    "rvll_set_wander_exact": (C.c_int, [Handle, C.c_int32]),
    "rvll_comm_unique_id": (C.c_int, [C.POINTER(C.c_ubyte)]),
    "rvll_comm_init": (C.c_int, [Handle, C.POINTER(C.c_ubyte), C.c_int32, C.c_int32]),
    "rvll_comm_add_lanes": (C.c_int, [Handle, C.c_int32, _ip]),
    "rvll_comm_set_lanes": (C.c_int, [Handle, C.c_int32]),
    "rvll_allgather_host": (C.c_int, [Handle, _dp, C.c_int64, _dp]),
    "rvll_runtime_info": (C.c_int, [C.c_char_p, C.c_int32]),
    "rvll_allgather_logl": (C.c_int, [Handle, C.c_int64]),
    "rvll_download_gathered": (C.c_int, [Handle, C.c_int64, _dp]),
    "rvll_allgather_theta": (C.c_int, [Handle, C.c_int64]),
    "rvll_download_gathered_theta": (C.c_int, [Handle, C.c_int64, _dp]),
    "rvll_comm_destroy": (C.c_int, [Handle]),
    "rvll_kep_rv_batch": (C.c_int, [Handle, _dp, C.c_int64, _dp, C.c_int32, C.c_uint32, _dp]),
    "rvll_fip_accumulate": (C.c_int, [C.c_int32, _dp, _dp, C.c_int32, _dp, _dp, C.POINTER(C.c_int64), C.c_int32,
                                      C.c_int32, _dp, C.c_int32, C.POINTER(FipTiming)]),
    "rvll_dev_trace_loglike": (C.c_int, [Handle, C.c_int64, C.c_int32, C.POINTER(C.c_uint64), C.c_int64, _ip, _ip]),
    "rvll_debug_eval": (C.c_int, [Handle, C.c_int32, _dp, _dp, C.c_int64, _dp]),
    "rvll_last_error": (C.c_char_p, []),
    "rvll_version": (C.c_int, [_ip, _ip]),
    "rvll_device_count": (C.c_int, [_ip]),
    "rvll_device_name": (C.c_int, [C.c_int32, C.c_char_p, C.c_int32]),
}


class RvllLibraryError(RuntimeError):
    """librvll.so is missing or cannot be loaded — the product has no fallback."""


class RvllError(RuntimeError):
    """A C-ABI call returned a negative RVLL_E_* code."""

    def __init__(self, code, message):
        self.code = code
        super().__init__(f"{ERROR_NAMES.get(code, code)}: {message}")


LIB_PATH = Path(__file__).resolve().parent / "librvll.so"
_lib = None


HW_QUEUES_NOTE = None          # what load() did about GPU_MAX_HW_QUEUES, and whether it can have taken effect (bench.py prints it)


def _hip_runtime_loaded():
    """Is a libamdhip64 already mapped into this process (torch imported first, a profiler's preload, ...)?"""
    try:
        with open("/proc/self/maps") as maps:
            return any("libamdhip64" in line for line in maps)
    except OSError:
        return None


def load():
    """Load the in-tree HIP library (once) and bind every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("RVLL_LIBRARY", LIB_PATH))
    if not path.exists():
        raise RvllLibraryError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C evidence_amd/csrc`. evidence_amd has no CPU fallback.")
    # The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and reads the variable when it
    # starts.  A handle has seven streams; when a copy stream of the streamed host batches shares the kernels' queue, every
    # chunk's kernels wait for a download (csrc/rvll_api.hip, stream_reserve: 3.3 against 2.0 ms at 262144 rows).  Ask for 8
    # unless the user has chosen; results do not depend on it.
    global HW_QUEUES_NOTE
    hip_already = _hip_runtime_loaded()
    chosen = "GPU_MAX_HW_QUEUES" in os.environ
    if not os.environ.get("RVLL_KEEP_HW_QUEUES"):            # (measurement switch: leave the variable alone, here and in librvll's constructor)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    HW_QUEUES_NOTE = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "unset"), "set_by": "the caller" if chosen else "evidence_amd (default 8)",
                      "hip_runtime_loaded_before_librvll": hip_already,
                      "in_effect": "unknown: the HIP runtime was already up and reads the variable only when it starts" if hip_already and not chosen
                                   else os.environ.get("GPU_MAX_HW_QUEUES", "unset (runtime default: 4)")}
    try:
        lib = C.CDLL(str(path))
    except OSError as exc:
        raise RvllLibraryError(f"cannot load {path}: {exc}") from exc
    for name, (restype, argtypes) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise RvllLibraryError(f"{path} does not export {name}") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    major, minor = C.c_int32(-1), C.c_int32(-1)
    lib.rvll_version(C.byref(major), C.byref(minor))
    if (major.value, minor.value) != ABI_VERSION:      # signatures change with the minor version while it is 0.x
        raise RvllLibraryError(
            f"{path} is librvll {major.value}.{minor.value}, these bindings are for {ABI_VERSION[0]}.{ABI_VERSION[1]}: "
            f"rebuild it (make -C evidence_amd/csrc)")
    _lib = lib
    return lib


def check(rc):
    """Raise RvllError for a negative return code."""
    if rc != OK:
        msg = load().rvll_last_error()
        raise RvllError(rc, msg.decode("utf-8", "replace") if msg else "")
    return rc


def as_dp(arr):
    return arr.ctypes.data_as(_dp)


def as_ip(arr):
    return arr.ctypes.data_as(_ip)
