"""CPU oracle of the FIP periodogram accumulation — TEST INFRASTRUCTURE ONLY (never imported by evidence_amd/).

A numpy restatement of evidence/fip_criterion.py:231-236 (frequency grid and window), :265-270 (model
probabilities from the median evidences) and :305-338 (the accumulation loop), kept in the reference's own
loop order so that every floating-point subtraction happens in the same sequence.  Pinned by
tests/test_fip_oracle.py against periodograms the reference script itself wrote
(tests/golden/fip_*.npz, made by tests/golden/gen_fip_golden.py).  The C routine `rvo_fip_accumulate`
(oracle/rvll_oracle.c) is the same fold for large inputs and is pinned against this file and the same fixtures.
"""
import numpy as np
from scipy.special import logsumexp

NFREQ = 50000                 # fip_criterion.py:229
COEF_WINDOW = 1.0             # :230


def frequency_grid(pmin, pmax, tobs, nfreq=NFREQ, coef_window=COEF_WINDOW):
    nu = np.linspace(2 * np.pi / pmax, 2 * np.pi / pmin, nfreq)      # :233
    nu_window = coef_window * 2 * np.pi / tobs                       # :234
    return nu, nu - nu_window / 2, nu + nu_window / 2                # :235-236


def model_probabilities(logzs_per_run):
    """logzs_per_run[r][k] -> p(k | y) from the median over runs (:243-270)."""
    logzs = np.median(np.asarray(logzs_per_run, dtype=float), axis=0)
    return np.exp(logzs - logsumexp(logzs))


def accumulate(posteriors, pky, nua, nub):
    """posteriors[r][k] = (samples [n,k], weights [n]) for k >= 1 (entry 0 unused).  Returns fapnu [R, nfreq]."""
    nfreq = len(nua)
    fapnu = np.ones([len(posteriors), nfreq])                         # :307
    for run, per_k in enumerate(posteriors):                          # :310
        for kmod in range(1, len(per_k)):                             # :312
            samples, weights = per_k[kmod]
            weights = np.array(weights, dtype=float)
            weights /= np.sum(weights)                                # :315
            for i, x in enumerate(samples):                           # :319
                x_freqs = 2 * np.pi / x                               # :321
                beg = np.searchsorted(nub, x_freqs, 'right')          # :334
                end = np.searchsorted(nua, x_freqs, 'left')           # :335
                listind = []
                for bi, ei in zip(beg, end):
                    listind += range(bi, ei)
                fapnu[run, listind] -= pky[kmod] * weights[i]         # :339 (a repeated index is applied once)
    return fapnu


def summary(fapnu):
    """:347-388 — clipped log10, per-frequency spread across runs, median/std/mean."""
    cut = np.maximum(fapnu, 1e-15)
    log10fips = np.log10(cut)
    diffs = np.max(log10fips, axis=0) - np.min(log10fips, axis=0)
    return {"log10fips": log10fips, "diffs": diffs, "median": np.median(log10fips, axis=0),
            "std": np.std(log10fips, axis=0), "mean": np.mean(cut, axis=0)}
