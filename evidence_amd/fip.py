"""FIP periodogram (Hara et al. 2021) accumulation on the GPU — the data-parallel part of
evidence/fip_criterion.py.

The reference script walks a directory of finished runs, unpickles their posteriors and then runs, per
independent run, a Python loop over every posterior sample of every planet model that subtracts
p(k|y)·w_i from every frequency bin within half a window of one of the sample's orbital frequencies
(fip_criterion.py:305-339).  Here the loop is `rvll_fip_accumulate` (include/rvll.h): the host only flattens
the posteriors into rows in the reference's loop order; the result is bit-identical.

Not reproduced: the directory walk / pickle loading (:50-220, sampler-output specific), the `--with-alias`
branch (:322-332 reads `x_freqs` before assigning it — it raises NameError upstream) and the plots.
"""
import ctypes as C

import numpy as np
from scipy.special import logsumexp

from . import _abi

NFREQ = 50000                 # fip_criterion.py:229
COEF_WINDOW = 1.0             # :230


def frequency_grid(pmin, pmax, tobs, nfreq=NFREQ, coef_window=COEF_WINDOW):
    """nu, nua, nub of fip_criterion.py:233-236 (angular frequencies, rad/day)."""
    nu = np.linspace(2 * np.pi / pmax, 2 * np.pi / pmin, nfreq)
    nu_window = coef_window * 2 * np.pi / tobs
    return nu, nu - nu_window / 2, nu + nu_window / 2


def observation_span(datadict):
    """Tobs of :206-222: max - min over all instruments of the `rjd` (else `jdb`) column."""
    lo, hi = 1e9, 0.0
    for inst in datadict.values():
        data = inst["data"]
        t = data["rjd"].values if "rjd" in data else data["jdb"].values
        lo, hi = min(lo, float(np.min(t))), max(hi, float(np.max(t)))
    return hi - lo


def model_probabilities(logzs_per_run):
    """p(k | y) from the per-model median of ln Z over the runs (:243-270).  logzs_per_run[r][k]."""
    logzs = np.median(np.asarray(logzs_per_run, dtype=float), axis=0)
    return np.exp(logzs - logsumexp(logzs))


def flatten_posteriors(posteriors, pky):
    """posteriors[r][k] = (samples [n, k], weights [n]) for k >= 1 (entry 0 — the no-planet model — is
    ignored, it has no periods) -> periods [rows, np_max] NaN-padded, contrib [rows], run_start [R + 1],
    in the reference's loop order: run, then kmod, then sample (:310-319)."""
    np_max = max((np.atleast_2d(p[0]).shape[1] for per_k in posteriors for p in per_k[1:] if p is not None),
                 default=1)
    if np_max > _abi.FIP_MAX_PLANETS:
        raise ValueError(f"more than {_abi.FIP_MAX_PLANETS} periods per sample")
    entries, run_start, rows = [], [0], 0
    for per_k in posteriors:
        for kmod in range(1, len(per_k)):
            if per_k[kmod] is None:
                continue
            samples, weights = per_k[kmod]
            samples = np.asarray(samples, dtype=np.float64)
            samples = samples.reshape(len(samples), -1)
            if len(weights) != len(samples):
                raise ValueError("one weight per sample required")
            entries.append((rows, kmod, samples, weights))
            rows += len(samples)
        run_start.append(rows)
    periods = np.empty((rows, np_max))
    contrib = np.empty(rows)
    for lo, kmod, samples, weights in entries:
        hi, k = lo + len(samples), samples.shape[1]
        periods[lo:hi, :k] = samples
        periods[lo:hi, k:] = np.nan
        np.divide(weights, np.sum(weights), out=contrib[lo:hi])              # :315  weights /= np.sum(weights)
        np.multiply(pky[kmod], contrib[lo:hi], out=contrib[lo:hi])            # :339  pky[kmod]*weights[i]
    return periods, contrib, np.asarray(run_start, dtype=np.int64)


def fip_periodogram(posteriors, pky, nua, nub, device=-1, repeats=1, return_timing=False):
    """fapnu [R, nfreq]: 1 - sum over models and samples of p(k|y) w_i [bin within the window of a sample
    frequency], accumulated on the GPU in the reference's order (bit-identical to :305-339)."""
    lib = _abi.load()
    nua = np.ascontiguousarray(nua, dtype=np.float64)
    nub = np.ascontiguousarray(nub, dtype=np.float64)
    if nua.shape != nub.shape or nua.ndim != 1:
        raise ValueError("nua and nub must be 1-D arrays of one length")
    periods, contrib, run_start = flatten_posteriors(posteriors, np.asarray(pky, dtype=np.float64))
    fapnu = np.ones((len(posteriors), nua.size))                              # :307
    timing = _abi.FipTiming()
    _abi.check(lib.rvll_fip_accumulate(
        int(device), _abi.as_dp(nua), _abi.as_dp(nub), nua.size, _abi.as_dp(periods), _abi.as_dp(contrib),
        run_start.ctypes.data_as(C.POINTER(C.c_int64)), len(posteriors), periods.shape[1], _abi.as_dp(fapnu),
        int(repeats), C.byref(timing)))
    if return_timing:
        return fapnu, {"index_ms": timing.index_ms, "accumulate_ms": timing.accumulate_ms, "rows": timing.rows,
                       "repeats": timing.repeats}
    return fapnu


def fip_summary(fapnu, nu):
    """The statistics the script derives from the matrix (:347-388): clipped log10 FIP per run, the
    convergence test (max - min over runs > 1 dex), median / std over runs and the mean FIP."""
    cut = np.maximum(fapnu, 1e-15)
    log10fips = np.log10(cut)
    diffs = np.max(log10fips, axis=0) - np.min(log10fips, axis=0)
    failed = np.where(diffs > 1)[0]
    return {"log10fips": log10fips, "diffs": diffs, "converged": failed.size == 0,
            "failed_periods": 2 * np.pi / np.asarray(nu)[failed],
            "median": np.median(log10fips, axis=0), "std": np.std(log10fips, axis=0),
            "mean": np.mean(cut, axis=0), "periods": 2 * np.pi / np.asarray(nu)}
