"""ctypes front end of oracle/librvll_oracle.so (the C restatement of the reference's
RV log-likelihood) and of oracle/_ref/libtrueanomaly_ref.so (the reference's own
Kepler solver compiled from /root/reference by `make -C oracle ref`).

TEST INFRASTRUCTURE ONLY — never imported by evidence_amd/.
"""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from evidence_amd import _abi
from evidence_amd.layout import ModelLayout

HERE = Path(__file__).resolve().parent
LIB = HERE / "librvll_oracle.so"
REF_LIB = HERE / "_ref" / "libtrueanomaly_ref.so"
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lib = None
_ref = None


def build(ref=True):
    """Compile the oracle (and, when /root/reference exists, the reference solver)."""
    subprocess.run(["make", "-s", "-C", str(HERE)], check=True)
    if ref and Path("/root/reference/evidence/rvmodel/trueanomaly.c").exists():
        subprocess.run(["make", "-s", "-C", str(HERE), "ref"], check=True)


def load():
    global _lib
    if _lib is None:
        if not LIB.exists():
            build(ref=False)
        lib = C.CDLL(str(LIB))
        lib.rvo_trueanomaly.restype = C.c_int
        lib.rvo_trueanomaly.argtypes = [_dp, C.c_int, C.c_double, _dp, C.c_int, C.c_double, _ip]
        lib.rvo_loglike_batch.restype = C.c_int
        lib.rvo_loglike_batch.argtypes = [C.POINTER(_abi.Layout), _dp, _dp, _dp, _ip, C.c_int, _dp, _dp,
                                          C.c_long, _dp, _ip, C.c_int]
        lib.rvo_iteration_counts.restype = C.c_int
        lib.rvo_iteration_counts.argtypes = [C.POINTER(_abi.Layout), _dp, C.c_int, _dp, _ip]
        lib.rvo_max_threads.restype = C.c_int
        lib.rvo_set_trig_perturb.restype = None
        lib.rvo_set_trig_perturb.argtypes = [C.c_double]
        lib.rvo_kep_rv_batch.restype = C.c_int
        lib.rvo_kep_rv_batch.argtypes = [C.POINTER(_abi.Layout), _dp, C.c_long, _dp, C.c_int, C.c_uint, _dp]
        lib.rvo_fip_accumulate.argtypes = [_dp, _dp, C.c_int, _dp, _dp, C.c_long, C.c_int, _dp]
        lib.rvo_fip_accumulate.restype = C.c_int
        _lib = lib
    return _lib


def load_ref():
    """The reference's own trueanomaly(), or None when oracle/_ref was not built."""
    global _ref
    if _ref is None and REF_LIB.exists():
        lib = C.CDLL(str(REF_LIB))
        lib.trueanomaly.restype = C.c_int
        lib.trueanomaly.argtypes = [_dp, C.c_int, C.c_double, _dp, C.c_int, C.c_double]
        _ref = lib
    return _ref


def trueanomaly(M, ecc, itmax=10000, tol=1e-4, want_iters=False):
    M = np.ascontiguousarray(M, dtype=np.float64)
    nu = np.zeros_like(M)
    iters = np.zeros(M.shape[0], dtype=np.int32)
    rc = load().rvo_trueanomaly(_abi.as_dp(M), M.shape[0], float(ecc), _abi.as_dp(nu), int(itmax), float(tol),
                                _abi.as_ip(iters))
    return (nu, rc, iters) if want_iters else (nu, rc)


def ref_trueanomaly(M, ecc, itmax=10000, tol=1e-4):
    lib = load_ref()
    if lib is None:
        raise FileNotFoundError(str(REF_LIB))
    M = np.ascontiguousarray(M, dtype=np.float64)
    nu = np.zeros_like(M)
    rc = lib.trueanomaly(_abi.as_dp(M), M.shape[0], float(ecc), _abi.as_dp(nu), int(itmax), float(tol))
    return nu, rc


class OracleModel:
    """CPU evaluation of a compiled ModelLayout over an EpochTable."""

    def __init__(self, layout: ModelLayout, table, linpar_series=None):
        self.layout, self.table = layout, table
        self._c, self._keep = layout.to_c()
        self._series = None if linpar_series is None else np.ascontiguousarray(linpar_series, dtype=np.float64)
        self.lib = load()

    def loglike(self, theta, nthreads=1, return_flags=False):
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        if theta.ndim == 1:
            theta = theta.reshape(1, -1)
        assert theta.shape[1] == self.layout.ndim
        n = theta.shape[0]
        out = np.empty(n, dtype=np.float64)
        flags = np.zeros(n, dtype=np.int32)
        t = self.table
        rc = self.lib.rvo_loglike_batch(
            C.byref(self._c), _abi.as_dp(t.time), _abi.as_dp(t.vrad), _abi.as_dp(t.svrad), _abi.as_ip(t.inst_id),
            t.n_epochs, _abi.as_dp(self._series) if self._series is not None else None,
            _abi.as_dp(theta), n, _abi.as_dp(out), _abi.as_ip(flags), int(nthreads))
        if rc != 0:
            raise MemoryError("oracle allocation failed")
        return (out, flags) if return_flags else out

    def conditioning(self, theta, nthreads=1, eps=2.0 ** -53):
        """How far log-L moves (relative) when sin / cos inside the Newton loop are nudged by one unit in the last place
        either way (rvo_set_trig_perturb): per row, max over the two signs.  Rows where this exceeds the parity bar are
        rows whose reference value is an accident of the reference's libm."""
        base = self.loglike(theta, nthreads)
        worst = np.zeros_like(base)
        try:
            for p in (eps, -eps):
                self.lib.rvo_set_trig_perturb(p)
                got = self.loglike(theta, nthreads)
                with np.errstate(divide="ignore", invalid="ignore"):
                    d = np.where(got == base, 0.0, np.abs(got - base) / np.maximum(np.abs(base), 1e-300))
                worst = np.maximum(worst, d)
        finally:
            self.lib.rvo_set_trig_perturb(0.0)
        return worst

    def kep_rv(self, theta, times, include_mask):
        theta = np.ascontiguousarray(np.atleast_2d(theta), dtype=np.float64)
        times = np.ascontiguousarray(times, dtype=np.float64)
        out = np.empty((theta.shape[0], times.shape[0]))
        rc = self.lib.rvo_kep_rv_batch(C.byref(self._c), _abi.as_dp(theta), theta.shape[0], _abi.as_dp(times),
                                       times.shape[0], int(include_mask), _abi.as_dp(out))
        if rc != 0:
            raise MemoryError("oracle allocation failed")
        return out

    def iteration_counts(self, theta):
        theta = np.ascontiguousarray(theta, dtype=np.float64).reshape(-1)
        it = np.zeros((max(1, self.layout.nplanets), self.table.n_epochs), dtype=np.int32)
        self.lib.rvo_iteration_counts(C.byref(self._c), _abi.as_dp(self.table.time), self.table.n_epochs,
                                      _abi.as_dp(theta), _abi.as_ip(it))
        return it[: self.layout.nplanets]


def max_threads():
    return int(load().rvo_max_threads())


def fip_accumulate(nua, nub, periods, contrib, run_start):
    """The C fold of rvo_fip_accumulate, run by run: fapnu [R, nfreq] starting from ones."""
    lib = load()
    nua = np.ascontiguousarray(nua, dtype=np.float64)
    nub = np.ascontiguousarray(nub, dtype=np.float64)
    periods = np.ascontiguousarray(periods, dtype=np.float64)
    contrib = np.ascontiguousarray(contrib, dtype=np.float64)
    fapnu = np.ones((len(run_start) - 1, nua.size))
    for r in range(len(run_start) - 1):
        lo, hi = int(run_start[r]), int(run_start[r + 1])
        row = fapnu[r]
        p = np.ascontiguousarray(periods[lo:hi])
        c = np.ascontiguousarray(contrib[lo:hi])
        rc = lib.rvo_fip_accumulate(nua.ctypes.data_as(_dp), nub.ctypes.data_as(_dp), nua.size,
                                    p.ctypes.data_as(_dp), c.ctypes.data_as(_dp), hi - lo, periods.shape[1],
                                    row.ctypes.data_as(_dp))
        if rc:
            raise RuntimeError("rvo_fip_accumulate failed")
    return fapnu
