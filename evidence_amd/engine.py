"""GpuRVModel — the host-side mirror of the reference's RVModel for the hot path.

Duck-types what the reference's sampler wrappers use from a model
(evidence/polychord/__init__.py:92,100-102,216-218; evidence/ultranest/__init__.py:93):
`.parnames` (sorted), `.log_likelihood(x)`, `.datadict`, `.fixedpardict`, `.nplanets`,
plus the batched forms the GPU exists for.  Every evaluation goes through the C-ABI
(include/rvll.h) into the HIP kernels; there is no host implementation.
"""
import ctypes as C
import threading
import weakref
from typing import Dict, Optional, Sequence

import numpy as np

from . import _abi
from .data import EpochTable
from .layout import compile_layout
from .priors import PriorSpec


# ---- large result arrays ---------------------------------------------------------------------------------
# A float64 array above glibc's 32 MB mmap threshold is a fresh mapping every time it is made and an munmap when it dies:
# 2.7 ms of page faults for the 40 MB of theta rows a 262144-point call returns — more than the call itself (1.8 ms;
# profiles/r03_stream_probe.txt).  Results of that size are therefore handed out from a small store of blocks that come back
# when the LAST array looking at them (the result or any view of it) is collected: the array's base is a lease object, numpy
# keeps it alive exactly as long as the memory is referenced, and its finalizer returns the block.
_RESULT_MIN_BYTES = 32 << 20
_RESULT_KEEP = 4                      # blocks kept for reuse (of any size); more are simply freed
_result_blocks = []                   # uint8 arrays
_result_lock = threading.RLock()      # (re-entrant: a lease's finalizer may run inside _result_array, at any allocation)


class _Lease:
    __slots__ = ("__array_interface__", "__weakref__")

    def __init__(self, block, shape):
        iface = dict(block.__array_interface__)
        iface.update(shape=tuple(shape), typestr="<f8", descr=[("", "<f8")], strides=None)
        self.__array_interface__ = iface


def _give_back(block):
    with _result_lock:
        if len(_result_blocks) < _RESULT_KEEP:
            _result_blocks.append(block)


def _result_array(shape):
    """An uninitialised C-contiguous float64 array of `shape` (np.empty below 32 MB)."""
    nbytes = 8 * int(np.prod(shape))
    if nbytes < _RESULT_MIN_BYTES:
        return np.empty(shape, dtype=np.float64)
    block = None
    with _result_lock:
        for i, b in enumerate(_result_blocks):
            if b.nbytes == nbytes:
                block = _result_blocks.pop(i)
                break
        if block is None and len(_result_blocks) >= _RESULT_KEEP:
            _result_blocks.pop(0)     # make room: the oldest block of another size goes
    if block is None:
        block = np.empty(nbytes, dtype=np.uint8)
    lease = _Lease(block, shape)
    weakref.finalize(lease, _give_back, block)
    return np.asarray(lease)


class GpuRVModel:
    """RV model with its epoch table resident on one MI355X.

    Parameters follow RVModel.__init__ (evidence/rvmodel/__init__.py:94-154):
    fixedpardict  {name: value} of fixed parameters
    datadict      {instrument: {'data': table with rjd|jdb, vrad, svrad}}  (or an EpochTable)
    parnames      names of the free parameters; theta is ordered as sorted(parnames)
    linpar_dict   optional {key: series[Ne]} for `linpar_{key}` terms (rvmodel:131-136,210-212)
    priordict     optional {parname: PriorSpec}; enables prior_transform*
    device        HIP device index (default: current device)
    tol, itmax    Newton stop rule of the Kepler solver; defaults 1e-4 and 10000 are what the
                  reference passes (rvmodel/__init__.py:466,491) — change them and parity is gone
    precision     "fp64" (the reference's arithmetic; the only parity mode), "mixed" (fp64 phase and
                  chi^2, fp32 Newton iteration) or "fp32" — BASELINE.json configs[4] tolerance sweep
    """

    def __init__(self, fixedpardict: Dict[str, float], datadict, parnames: Sequence[str],
                 linpar_dict: Optional[Dict[str, np.ndarray]] = None,
                 priordict: Optional[Dict[str, PriorSpec]] = None, device: int = -1,
                 tol: Optional[float] = None, itmax: Optional[int] = None, precision: str = "fp64"):
        self._lib = _abi.load()          # raises RvllLibraryError when the HIP library is absent
        self._h = _abi.Handle()
        self.fixedpardict = dict(fixedpardict)
        self.table = datadict if isinstance(datadict, EpochTable) else EpochTable.from_datadict(datadict)
        self.datadict = datadict if not isinstance(datadict, EpochTable) else self.table.to_datadict()
        self.insts = list(self.table.insts)
        self.linpar_dict = dict(linpar_dict) if linpar_dict else {}
        self.layout = compile_layout(parnames, self.fixedpardict, self.insts, list(self.linpar_dict))
        if tol is not None:
            self.layout.tol = float(tol)
        if itmax is not None:
            self.layout.itmax = int(itmax)
        if precision not in _abi.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_abi.PRECISIONS)}")
        self.precision = precision
        self.layout.precision = _abi.PRECISIONS[precision]
        self.parnames = list(self.layout.parnames)
        self.nplanets = self.layout.nplanets
        self.drift_in_model = self.layout.has_drift
        self.linpar_in_model = self.layout.has_linpar
        self.jitter_in_model = self.layout.has_jitter
        self.time, self.vrad, self.svrad = self.table.time, self.table.vrad, self.table.svrad
        self.model_path = None

        series = None
        if self.layout.linpar_names:
            series = np.ascontiguousarray(
                np.stack([np.asarray(self.linpar_dict[k], dtype=np.float64) for k in self.layout.linpar_names]))
            if series.shape != (len(self.layout.linpar_names), self.table.n_epochs):
                raise ValueError("each linpar series must have one value per epoch")
        self._series = series
        layout_c, self._layout_keep = self.layout.to_c()
        _abi.check(self._lib.rvll_create(
            C.byref(layout_c), _abi.as_dp(self.table.time), _abi.as_dp(self.table.vrad),
            _abi.as_dp(self.table.svrad), _abi.as_ip(self.table.inst_id), self.table.n_epochs,
            _abi.as_dp(series) if series is not None else None, int(device), C.byref(self._h)))
        # persistent buffers (and their ctypes pointers) of the scalar callbacks: per call only the copy of x
        # and one foreign call remain on the Python side
        nd = max(1, self.ndim)
        self._s_in, self._s_th = np.empty((1, nd)), np.empty((1, nd))
        self._s_out, self._s_flag = np.empty(1), np.zeros(1, dtype=np.int32)
        self._s_in_p, self._s_th_p = _abi.as_dp(self._s_in), _abi.as_dp(self._s_th)
        self._s_out_p, self._s_flag_p = _abi.as_dp(self._s_out), _abi.as_ip(self._s_flag)
        self.priordict = None
        self._live_n = 0                       # rows of the resident live set (live_init)
        if priordict is not None:
            self.set_priors(priordict)

    # ---- lifetime ---------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.rvll_destroy(self._h)
            self._h = _abi.Handle()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def ndim(self):
        return len(self.parnames)

    # ---- priors -------------------------------------------------------------------------
    def set_priors(self, priordict: Dict[str, PriorSpec]):
        """priordict[name] for every free parameter (evidence/polychord/__init__.py:152)."""
        specs = []
        for name in self.parnames:
            if name not in priordict:
                raise KeyError(name)
            specs.append(priordict[name])
        arr = (_abi.Prior * max(1, len(specs)))()
        for i, s in enumerate(specs):
            arr[i] = s.to_c()
        _abi.check(self._lib.rvll_set_priors(self._h, arr, len(specs)))
        self.priordict = dict(priordict)

    # ---- log-likelihood --------------------------------------------------------------------
    def _theta2d(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.ndim != 2 or x.shape[1] != self.ndim:
            raise ValueError(f"expected an array of shape (n, {self.ndim}), got {x.shape}")
        return x

    @staticmethod
    def _out_array(a, shape, name):
        if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous and a.flags.writeable
                and a.shape == tuple(shape)):
            raise ValueError(f"{name}: expected a writeable C-contiguous float64 array of shape {tuple(shape)}")
        return a

    def log_likelihood_batch(self, X, return_flags=False):
        """log-L of every row of X[n, ndim] -> float64[n] (one fused kernel launch)."""
        X = self._theta2d(X)
        n = X.shape[0]
        out = np.empty(n, dtype=np.float64)
        flags = np.zeros(n, dtype=np.int32)
        _abi.check(self._lib.rvll_loglike_batch(self._h, _abi.as_dp(X), n, _abi.as_dp(out), _abi.as_ip(flags)))
        return (out, flags) if return_flags else out

    def log_likelihood(self, x):
        """Scalar form, same signature as RVModel.log_likelihood (rvmodel/__init__.py:157)."""
        x = np.asarray(x, dtype=np.float64)
        if x.size != self.ndim:
            raise ValueError(f"expected {self.ndim} parameters, got {x.size}")
        self._s_in[0, :self.ndim] = x.ravel()
        rc = self._lib.rvll_loglike_batch(self._h, self._s_in_p, 1, self._s_out_p, self._s_flag_p)
        if rc:
            _abi.check(rc)
        return float(self._s_out[0])

    # ---- Keplerian curves for post-processing -----------------------------------------------------
    def _curves(self, X, time, mask):
        X = self._theta2d(np.atleast_2d(np.asarray(X, dtype=np.float64)))
        time = np.ascontiguousarray(np.atleast_1d(time), dtype=np.float64)
        out = np.empty((X.shape[0], time.shape[0]), dtype=np.float64)
        _abi.check(self._lib.rvll_kep_rv_batch(self._h, _abi.as_dp(X), X.shape[0], _abi.as_dp(time), time.shape[0],
                                               int(mask), _abi.as_dp(out)))
        return out

    def kep_rv_batch(self, X, time, exclude_planet=None):
        """RVModel.kep_rv for every row of X: the summed Keplerian RV of all planets except
        `exclude_planet` (1-based, as rvmodel/__init__.py:343-385) at `time` -> [n, len(time)]."""
        if not (exclude_planet is None or type(exclude_planet) is int):
            raise AssertionError(f"exclude_planet has to be an {int}, got {type(exclude_planet)}.")   # rvmodel:365-366
        mask = (1 << self.nplanets) - 1
        if exclude_planet is not None and 1 <= exclude_planet <= self.nplanets:
            mask &= ~(1 << (exclude_planet - 1))
        return self._curves(X, time, mask)

    def modelk_batch(self, X, time, planet):
        """RVModel.modelk for every row of X: the Keplerian curve of one planet (1-based, rvmodel:388-463)."""
        if not 1 <= int(planet) <= self.nplanets:
            raise KeyError(f"planet{planet}_k1")
        return self._curves(X, time, 1 << (int(planet) - 1))

    # ---- prior transform --------------------------------------------------------------------
    def prior_transform_batch(self, cubes):
        cubes = self._theta2d(cubes)
        out = _result_array(cubes.shape)
        _abi.check(self._lib.rvll_prior_batch(self._h, _abi.as_dp(cubes), cubes.shape[0], _abi.as_dp(out)))
        return out

    def prior_transform(self, cube):
        cube = np.asarray(cube, dtype=np.float64)
        if cube.size != self.ndim:
            raise ValueError(f"expected {self.ndim} coordinates, got {cube.size}")
        self._s_in[0, :self.ndim] = cube.ravel()
        rc = self._lib.rvll_prior_batch(self._h, self._s_in_p, 1, self._s_th_p)
        if rc:
            _abi.check(rc)
        return self._s_th[0, :self.ndim].copy().reshape(cube.shape)

    def prior_loglike(self, cube):
        """Scalar pair: prior(cube) and the log-L of the result in ONE call — (theta, logL).  With the scalar server on,
        one request of the persistent kernel instead of two (what PolyChord's prior + loglike sequence amounts to)."""
        cube = np.asarray(cube, dtype=np.float64)
        if cube.size != self.ndim:
            raise ValueError(f"expected {self.ndim} coordinates, got {cube.size}")
        self._s_in[0, :self.ndim] = cube.ravel()
        rc = self._lib.rvll_prior_loglike_batch(self._h, self._s_in_p, 1, self._s_th_p, self._s_out_p, self._s_flag_p)
        if rc:
            _abi.check(rc)
        return self._s_th[0, :self.ndim].copy().reshape(cube.shape), float(self._s_out[0])

    def prior_loglike_batch(self, cubes, return_flags=False, theta_out=None, logl_out=None):
        """Fused prior(cube) -> theta -> log-L: one upload, two launches, one download.

        theta_out / logl_out: arrays to fill (C-contiguous float64, (n, ndim) and (n,)) — a caller that evaluates hundreds of
        thousands of rows per call keeps them between calls: above glibc's 32 MB mmap threshold a fresh result array is a
        fresh mapping, and its page faults cost more than the whole call (profiles/r03_stream_probe.txt)."""
        cubes = self._theta2d(cubes)
        n = cubes.shape[0]
        theta = _result_array((n, self.ndim)) if theta_out is None else self._out_array(theta_out, (n, self.ndim), "theta_out")
        out = np.empty(n, dtype=np.float64) if logl_out is None else self._out_array(logl_out, (n,), "logl_out")
        flags = np.zeros(n, dtype=np.int32)
        _abi.check(self._lib.rvll_prior_loglike_batch(self._h, _abi.as_dp(cubes), n, _abi.as_dp(theta),
                                                      _abi.as_dp(out), _abi.as_ip(flags)))
        return (theta, out, flags) if return_flags else (theta, out)

    # ---- device-resident forms (bench / multi-GPU) ----------------------------------------------
    def dev_reserve(self, n):
        _abi.check(self._lib.rvll_dev_reserve(self._h, int(n)))

    def dev_upload_theta(self, X):
        X = self._theta2d(X)
        _abi.check(self._lib.rvll_dev_upload_theta(self._h, _abi.as_dp(X), X.shape[0]))

    def dev_upload_cube(self, cubes):
        cubes = self._theta2d(cubes)
        _abi.check(self._lib.rvll_dev_upload_cube(self._h, _abi.as_dp(cubes), cubes.shape[0]))

    def dev_fill_cube(self, n, seed):
        _abi.check(self._lib.rvll_dev_fill_cube(self._h, int(n), int(seed)))

    def dev_prior(self, n):
        _abi.check(self._lib.rvll_dev_prior(self._h, int(n)))

    def dev_loglike(self, n):
        _abi.check(self._lib.rvll_dev_loglike(self._h, int(n)))

    def slice_walk(self, cube, theta, logl, lstar, chol, wrapped=None, nsteps=10, max_rounds=200, seed=0,
                   walker_base=0):
        """nsteps slice-sampling moves of every walker inside logL > lstar, entirely on the GPU
        (rvll_slice_walk).  cube/theta/logl are the walkers' start points; returns (cube, theta, logl, ncalls)
        of the end points.  chol: lower-triangular factor of the live points' covariance in the unit cube.
        walker_base: index of the first row in a larger set of walkers (sharded walks draw the same random numbers
        as the unsharded one)."""
        cube = np.array(self._theta2d(cube), dtype=np.float64, order="C")
        theta = np.array(self._theta2d(theta), dtype=np.float64, order="C")
        logl = np.array(logl, dtype=np.float64).reshape(-1)
        k = cube.shape[0]
        if theta.shape != cube.shape or logl.shape[0] != k:
            raise ValueError("cube, theta and logl must describe the same walkers")
        chol = np.ascontiguousarray(chol, dtype=np.float64)
        if chol.shape != (self.ndim, self.ndim):
            raise ValueError("chol must be [ndim, ndim]")
        wr = None if wrapped is None else np.ascontiguousarray(np.asarray(wrapped, dtype=bool).astype(np.int32))
        ncalls = C.c_int64(0)
        _abi.check(self._lib.rvll_slice_walk(
            self._h, _abi.as_dp(cube), _abi.as_dp(theta), _abi.as_dp(logl), k, float(lstar), _abi.as_dp(chol),
            _abi.as_ip(wr) if wr is not None else None, int(nsteps), int(max_rounds), int(seed) & (2 ** 64 - 1),
            int(walker_base), C.byref(ncalls)))
        return cube, theta, logl, int(ncalls.value)

    # ---- live set resident on the device (nested.run_nested_slice(..., live=model)) -----------------------------
    def live_init(self, cube):
        """N unit-cube rows -> prior transform -> log-L; the live set stays on the device.  Returns log-L [N]."""
        cube = self._theta2d(cube)
        logl = np.empty(cube.shape[0], dtype=np.float64)
        _abi.check(self._lib.rvll_live_init(self._h, _abi.as_dp(cube), cube.shape[0], _abi.as_dp(logl)))
        self._live_n = cube.shape[0]
        return logl

    def live_step(self, order, kdead, start, lstar, wrapped=None, nsteps=10, max_rounds=200, seed=0, walker_base=0,
                  chol=None, return_chol=False):
        """One iteration of nested sampling on the resident live set (include/rvll.h, rvll_live_step): rows order[:kdead]
        die (kept in the device's dead store), kdead walkers start from rows `start`, walk, and replace them.  chol=None:
        the whitening comes from the surviving rows' covariance, computed on the device.  Returns (logl_new[kdead], ncalls)
        (+ the factor used with return_chol)."""
        order = None if order is None else np.ascontiguousarray(order, dtype=np.int32)     # None: the order live_sort left on the device
        start = np.ascontiguousarray(start, dtype=np.int32)
        kdead = int(kdead)
        if self._live_n < 1:
            raise RuntimeError("live_init has not been called")
        if (order is not None and order.shape != (self._live_n,)) or start.shape != (kdead,):
            raise ValueError("order must list every live row, start one row per dying point")
        wr = None if wrapped is None else np.ascontiguousarray(np.asarray(wrapped, dtype=bool).astype(np.int32))
        ch = None if chol is None else np.ascontiguousarray(chol, dtype=np.float64)
        if ch is not None and ch.shape != (self.ndim, self.ndim):
            raise ValueError("chol must be [ndim, ndim]")
        logl_new = np.empty(kdead, dtype=np.float64)
        used = np.empty((self.ndim, self.ndim)) if return_chol else None
        ncalls = C.c_int64(0)
        _abi.check(self._lib.rvll_live_step(
            self._h, _abi.as_ip(order) if order is not None else None, kdead, _abi.as_ip(start), float(lstar), _abi.as_dp(ch) if ch is not None else None,
            _abi.as_ip(wr) if wr is not None else None, int(nsteps), int(max_rounds), int(seed) & (2 ** 64 - 1),
            int(walker_base), C.byref(ncalls), _abi.as_dp(logl_new), _abi.as_dp(used) if used is not None else None))
        return (logl_new, int(ncalls.value), used) if return_chol else (logl_new, int(ncalls.value))

    def live_sort(self, kdead):
        """Sort the resident live points by log-L ON THE DEVICE (include/rvll.h, rvll_live_sort): returns (log-L of the kdead
        lowest in ascending order, lstar = the kdead-th lowest, the highest log-L).  The live_step that follows is called with
        order=None and `start` as ranks among the survivors."""
        kdead = int(kdead)
        if self._live_n < 1:
            raise RuntimeError("live_init has not been called")
        dead = np.empty(kdead, dtype=np.float64)
        lstar, top = C.c_double(), C.c_double()
        _abi.check(self._lib.rvll_live_sort(self._h, kdead, _abi.as_dp(dead), C.byref(lstar), C.byref(top)))
        return dead, lstar.value, top.value

    def live_get(self, cube=True, theta=True, logl=True, theta_out=None):
        """(cube, theta, logl) of the resident live set (None for the ones switched off); theta_out: a C-contiguous
        [n, ndim] float64 array (or view) to receive theta in place."""
        n = self._live_n
        u = np.empty((n, self.ndim)) if cube else None
        th = (theta_out if theta_out is not None else np.empty((n, self.ndim))) if theta else None
        ll = np.empty(n) if logl else None
        if th is not None and (th.shape != (n, self.ndim) or th.dtype != np.float64 or not th.flags.c_contiguous):
            raise ValueError("theta_out must be a C-contiguous float64 array of shape (n, ndim)")
        _abi.check(self._lib.rvll_live_get(self._h, _abi.as_dp(u) if cube else None, _abi.as_dp(th) if theta else None,
                                           _abi.as_dp(ll) if logl else None))
        return u, th, ll

    def live_dead_count(self):
        n = C.c_int64(0)
        _abi.check(self._lib.rvll_live_dead(self._h, C.byref(n), None, None))
        return int(n.value)

    def live_dead(self, theta_out=None):
        """(theta, logl) of every point that died so far, in the order they died; theta_out: a C-contiguous
        [n_dead, ndim] float64 array (or view) to receive theta in place."""
        n = self.live_dead_count()
        th = theta_out if theta_out is not None else np.empty((n, self.ndim))
        if th.shape != (n, self.ndim) or th.dtype != np.float64 or not th.flags.c_contiguous:
            raise ValueError("theta_out must be a C-contiguous float64 array of shape (n_dead, ndim)")
        ll = np.empty(n)
        if n:
            cnt = C.c_int64(n)
            _abi.check(self._lib.rvll_live_dead(self._h, C.byref(cnt), _abi.as_dp(th), _abi.as_dp(ll)))
        return th, ll

    def scalar_server(self, enable=True):
        """Answer scalar log_likelihood(x) calls through a persistent kernel polling pinned host memory (a PCIe
        round trip instead of a launch + synchronisation; same bits).  Any other call on this model stops the
        kernel first; it also leaves by itself after 5 ms without a request and restarts on the next one."""
        _abi.check(self._lib.rvll_scalar_server(self._h, 1 if enable else 0))

    def prior_table_info(self):
        """{parameter: (measured interpolation error, evaluated-by-interpolation flag)} for the Beta/Gamma priors."""
        out = {}
        for d, name in enumerate(self.parnames):
            err, direct = C.c_double(), C.c_int32()
            _abi.check(self._lib.rvll_prior_table_info(self._h, d, C.byref(err), C.byref(direct)))
            if err.value == err.value:
                out[name] = (err.value, bool(direct.value))
        return out

    def dev_prior_loglike(self, n):
        """cube -> theta -> log-L in one launch (the prior transform runs in the log-L kernel's staging step)."""
        _abi.check(self._lib.rvll_dev_prior_loglike(self._h, int(n)))

    def dev_flip_lane(self):
        """Alternate the pipeline lane between independent device-resident batches (include/rvll.h)."""
        return self._lib.rvll_dev_flip_lane(self._h)

    def dev_sync(self):
        _abi.check(self._lib.rvll_dev_sync(self._h))

    def dev_download(self, n, theta=False, logl=True, flags=False):
        n = int(n)
        th = _result_array((n, self.ndim)) if theta else None
        ll = np.empty(n, dtype=np.float64) if logl else None
        fl = np.empty(n, dtype=np.int32) if flags else None
        _abi.check(self._lib.rvll_dev_download(
            self._h, n, _abi.as_dp(th) if theta else None, _abi.as_dp(ll) if logl else None,
            _abi.as_ip(fl) if flags else None))
        return th, ll, fl

    def dev_mark(self, which):
        """Record HIP event 0 (start) or 1 (stop) on the compute stream."""
        _abi.check(self._lib.rvll_dev_mark(self._h, int(which)))

    def dev_mark_elapsed_ms(self):
        ms = C.c_double()
        _abi.check(self._lib.rvll_dev_mark_elapsed(self._h, C.byref(ms)))
        return ms.value

    def dev_time_loglike(self, n, warmup=3, iters=20):
        t = _abi.Timing()
        _abi.check(self._lib.rvll_dev_time_loglike(self._h, int(n), int(warmup), int(iters), C.byref(t)))
        return {f: getattr(t, f) for f, _ in _abi.Timing._fields_}

    def dev_trace_loglike(self, n, warmup=50):
        """Per-workgroup time stamps of one launch of the diagnostic twin of the fp64 kernel (include/rvll.h):
        returns (uint64[blocks, 8], points_per_block)."""
        nb, pb = C.c_int32(), C.c_int32()
        _abi.check(self._lib.rvll_dev_trace_loglike(self._h, int(n), 0, None, 0, C.byref(nb), C.byref(pb)))
        out = np.zeros((nb.value, 8), dtype=np.uint64)
        _abi.check(self._lib.rvll_dev_trace_loglike(self._h, int(n), int(warmup),
                                                    out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size,
                                                    C.byref(nb), C.byref(pb)))
        return out, pb.value

    def set_points_per_block(self, pb):
        _abi.check(self._lib.rvll_set_points_per_block(self._h, int(pb)))

    def set_kernel_form(self, form):
        """Launch form of the log-L kernel: "auto" (by batch size), "tile" or "cu" (include/rvll.h); same bits."""
        _abi.check(self._lib.rvll_set_kernel_form(self._h, {"auto": 0, "tile": 1, "cu": 2}[form]))

    def set_slim_table_range(self, umax):
        """Test hook (include/rvll.h): |logit q| range the table-only prior stage takes before handing over."""
        _abi.check(self._lib.rvll_set_slim_table_range(self._h, float(umax)))

    def set_wander_exact(self, on=True):
        """Redo wandering Kepler solves (e >= 0.97: more than eight Newton steps) with correctly rounded sin / cos (include/rvll.h,
        rvll_set_wander_exact; default on)."""
        _abi.check(self._lib.rvll_set_wander_exact(self._h, 1 if on else 0))

    def set_walk_speculation(self, max_ahead):
        """Candidates a walker of slice_walk may evaluate ahead per iteration in otherwise free tile slots (include/rvll.h;
        default 4, 1 = none).  Results do not depend on it."""
        _abi.check(self._lib.rvll_set_walk_speculation(self._h, int(max_ahead)))

    def slice_walk_evaluated(self):
        """Tile slots the last slice_walk evaluated (>= the calls it reported: unused speculative candidates included)."""
        n = C.c_int64(0)
        _abi.check(self._lib.rvll_slice_walk_evaluated(self._h, C.byref(n)))
        return int(n.value)

    def slice_walk_rounds(self):
        """Rounds the last slice_walk / live_step took in the rounds form (include/rvll.h); 0: a single-kernel form walked."""
        n = C.c_int32(0)
        _abi.check(self._lib.rvll_slice_walk_rounds(self._h, C.byref(n)))
        return int(n.value)

    def slice_walk_phases(self):
        """Diagnostic library build only (include/rvll.h): ticks per phase of the last slice_walk, summed over workgroups."""
        out = (C.c_uint64 * 6)()
        _abi.check(self._lib.rvll_slice_walk_phases(self._h, out))
        return [int(v) for v in out]

    def debug_eval(self, op, x, y=None):
        """Evaluate one device math routine elementwise (include/rvll.h, diagnostics)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        yy = None if y is None else np.ascontiguousarray(np.broadcast_to(y, x.shape), dtype=np.float64)
        out = np.empty_like(x)
        _abi.check(self._lib.rvll_debug_eval(self._h, int(op), _abi.as_dp(x), _abi.as_dp(yy) if yy is not None else None,
                                             x.size, _abi.as_dp(out)))
        return out

    # ---- multi-GPU ----------------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = (C.c_ubyte * _abi.COMM_ID_BYTES)()
        _abi.check(_abi.load().rvll_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, comm_id: bytes, nranks: int, rank: int):
        if len(comm_id) != _abi.COMM_ID_BYTES:
            raise ValueError("comm id must be 128 bytes")
        buf = (C.c_ubyte * _abi.COMM_ID_BYTES).from_buffer_copy(comm_id)
        _abi.check(self._lib.rvll_comm_init(self._h, buf, int(nranks), int(rank)))

    def comm_add_lanes(self, want=3):
        """Collective: try to add pipeline lanes (ncclCommSplit); returns how many this rank holds.  The ranks must
        agree on the minimum and call comm_set_lanes with it (include/rvll.h)."""
        have = C.c_int32(1)
        _abi.check(self._lib.rvll_comm_add_lanes(self._h, int(want), C.byref(have)))
        return have.value

    def comm_set_lanes(self, n):
        _abi.check(self._lib.rvll_comm_set_lanes(self._h, int(n)))

    def allgather_host(self, mine, world):
        """RCCL all-gather of a small float64 host array: [n] per rank -> [world, n] on every rank."""
        mine = np.ascontiguousarray(mine, dtype=np.float64).ravel()
        out = np.empty((int(world), mine.size), dtype=np.float64)
        _abi.check(self._lib.rvll_allgather_host(self._h, _abi.as_dp(mine), mine.size, _abi.as_dp(out)))
        return out

    @staticmethod
    def runtime_info():
        """Which HIP runtime / RCCL / librvll this process runs on (include/rvll.h rvll_runtime_info)."""
        import json
        buf = C.create_string_buffer(2048)
        _abi.check(_abi.load().rvll_runtime_info(buf, len(buf)))
        info = json.loads(buf.value.decode())
        info["hw_queues"] = _abi.HW_QUEUES_NOTE           # set by evidence_amd at load time? in effect?
        return info

    def allgather_logl(self, n_local):
        _abi.check(self._lib.rvll_allgather_logl(self._h, int(n_local)))

    def download_gathered(self, n_total):
        out = np.empty(int(n_total), dtype=np.float64)
        _abi.check(self._lib.rvll_download_gathered(self._h, int(n_total), _abi.as_dp(out)))
        return out

    def allgather_theta(self, n_local):
        """All-gather the resident theta rows (produced on the device by the prior transform) of every rank."""
        _abi.check(self._lib.rvll_allgather_theta(self._h, int(n_local)))

    def download_gathered_theta(self, n_total):
        out = np.empty((int(n_total), self.ndim), dtype=np.float64)
        _abi.check(self._lib.rvll_download_gathered_theta(self._h, int(n_total), _abi.as_dp(out)))
        return out

    def comm_destroy(self):
        _abi.check(self._lib.rvll_comm_destroy(self._h))
