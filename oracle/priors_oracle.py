"""CPU oracle of the prior transform — TEST INFRASTRUCTURE ONLY (never imported by evidence_amd/).

A numpy/scipy restatement of `.ppf` for every distribution of the reference, each citing the lines it
follows in evidence/priors.py.  scipy (pinned 1.15.3 in this image) is the reference's own dependency for
the special functions, so the Normal/LogNormal/Alpha/Beta/Gamma branches call the same `scipy.stats`
entry points the reference calls; the table-inverted families rebuild the reference's grid
(`arange(xmin, xmax + dx, dx)`, dx = (xmax - xmin) * 1e-4, priors.py:9-10) and invert it with
`numpy.interp`, which is what `scipy.interpolate.interp1d(cdf, x)` evaluates for 1-D linear tables.
Pinned by tests/test_priors_oracle.py against the golden vectors produced by the reference itself.
"""
import numpy as np
from scipy import special as sp
from scipy import stats

STEP = 1.0 / 1e4          # priors.py:9-10


def _grid(xmin, xmax):
    dx = (xmax - xmin) * STEP
    return np.arange(xmin, xmax + dx, dx)


def _interp_inverse(q, cdf, x, front_end, lo=-np.inf, hi=np.inf):
    """interp1d(cdf, x)(q) with interp1d's stable sort and bounds error (-> nan), optionally behind
    scipy's rv_continuous.ppf front end (q == 0 -> lower support, q == 1 -> upper support)."""
    order = np.argsort(cdf, kind="mergesort")
    cdf, x = cdf[order], x[order]
    q = np.asarray(q, dtype=float)
    out = np.full(q.shape, np.nan)
    inside = (q >= cdf[0]) & (q <= cdf[-1])
    if front_end:
        inside &= (q > 0) & (q < 1)
    out[inside] = np.interp(q[inside], cdf, x)
    if front_end:
        out[q == 0] = lo
        out[q == 1] = hi
    return out


def ppf(name, args, q):
    q = np.asarray(q, dtype=float)
    a = [float(v) for v in args]
    if name == "Uniform":                                   # priors.py:41-42
        return a[0] + (a[1] - a[0]) * q
    if name == "Jeffreys":                                  # :62-63
        return a[0] * (a[1] / a[0]) ** q
    if name == "ModJeffreys":                               # :82-83
        return a[0] * ((1 + a[1] / a[0]) ** q) - a[0]
    if name == "UniformFrequency":                          # :100-101
        return a[0] / (1 - q * (a[1] - a[0]) / a[1])
    if name == "TruncatedRayleigh":                         # :249-252
        A = 1 - np.exp(-a[1] ** 2 / (2 * a[0] ** 2))
        return np.sqrt(-2 * a[0] ** 2 * np.log(1 - (q * A)))
    # the scipy-backed families call the very entry points the reference does (priors.py:375-376, 397-398,
    # 424-425, 436-437): scipy.stats' own ppf, not a re-derivation
    with np.errstate(all="ignore"):
        if name == "Normal":
            return stats.norm(*a).ppf(q)
        if name == "LogNormal":
            return stats.lognorm(*a).ppf(q)
        if name == "Alpha":
            return stats.alpha.ppf(q, a[0])
        if name == "Beta":
            return stats.beta.ppf(q, a[0], a[1])
        if name == "Gamma":
            return stats.gamma.ppf(q, a[0], scale=1.0 / a[1])
    if name == "Binormal":                                  # :118-124
        mu1, s1, mu2, s2, A = a
        x = _grid(mu1 - 9.0 * s1, mu2 + 9.0 * s2)
        cdf = 0.5 * (sp.ndtr((x - mu1) / s1) * (1.0 - A) + sp.ndtr((x - mu2) / s2) * (1.0 + A))
        return _interp_inverse(q, cdf, x, True)
    if name == "AsymmetricNormal":                          # :183-202
        mu, s1, s2 = a
        x = _grid(mu - 9 * s1, mu + 9 * s2)
        k1, k2 = 2.0 * s1 / (s1 + s2), 2.0 * s2 / (s1 + s2)
        cdf = np.where(x <= mu, sp.ndtr((x - mu) / s1) * k1, k1 * 0.5 + (sp.ndtr((x - mu) / s2) - 0.5) * k2)
        return _interp_inverse(q, cdf, x, True)
    if name == "TruncatedUNormal":                          # :214-228 (ppf overridden: no front end)
        mu, sg, xmin, xmax = a
        x = _grid(xmin, xmax)
        lo = sp.ndtr((xmin - mu) / sg)
        cdf = (sp.ndtr((x - mu) / sg) - lo) / (sp.ndtr((xmax - mu) / sg) - lo)
        cdf = np.where(x >= xmin, cdf, 0.0)
        cdf = np.where(x < xmax, cdf, 1.0)
        return _interp_inverse(q, cdf, x, False)
    if name == "PowerLaw":                                  # :275-287
        al, xmin, xmax = a
        x = _grid(xmin, xmax)
        cdf = (x ** (1.0 + al) - xmin ** (1.0 + al)) / (xmax ** (1.0 + al) - xmin ** (1.0 + al))
        cdf = np.where(x > xmin, cdf, 0.0)
        cdf = np.where(x >= xmax, 1.0, cdf)
        return _interp_inverse(q, cdf, x, True)
    if name == "DoublePowerLaw":                            # :306-326
        al, be, x0, xmin, xmax = a
        x = _grid(xmin, xmax)
        with np.errstate(all="ignore"):
            a1 = (x0 ** (1.0 + al) - xmin ** (1.0 + al)) / (al + 1.0)
            a2 = (xmax ** (1.0 + be) - x0 ** (1.0 + be)) / (be + 1.0)
            join = x0 ** al / x0 ** be
            A = 1.0 / (a1 + join * a2)
            low = A * (x ** (1.0 + al) - xmin ** (1.0 + al)) / (1.0 + al)
            high = A * a1 + join * A * (x ** (1.0 + be) - x0 ** (1.0 + be)) / (1.0 + be)
        cdf = np.where(x < x0, low, high)
        cdf = np.where(x > xmin, cdf, 0.0)
        cdf = np.where(x >= xmax, 1.0, cdf)
        return _interp_inverse(q, cdf, x, True)
    if name == "Sine":                                      # :340-354 (support 0..180 degrees, :456-457)
        xmin, xmax = a
        x = _grid(xmin, xmax)
        lo, hi = max(xmin, 0.0), min(xmax, 180.0)
        rad = np.pi / 180.0
        cdf = (np.cos(lo * rad) - np.cos(x * rad)) / (np.cos(lo * rad) - np.cos(hi * rad))
        cdf = np.where(x >= lo, cdf, 0.0)
        cdf = np.where(x <= hi, cdf, 1.0)
        return _interp_inverse(q, cdf, x, True, 0.0, 180.0)
    raise KeyError(name)


def sorted_uniform(cube_group, a, b, log=False):
    """pypolychord's forced-identifiability transform (priors.py:462-467; source not in the checkout —
    parity unpinned): t[N-1] = x[N-1]^(1/N), t[n] = x[n]^(1/(n+1)) t[n+1], then a + (b-a) t or a (b/a)^t."""
    x = np.asarray(cube_group, dtype=float)
    n = x.shape[-1]
    t = np.empty_like(x)
    t[..., n - 1] = x[..., n - 1] ** (1.0 / n)
    for k in range(n - 2, -1, -1):
        t[..., k] = x[..., k] ** (1.0 / (k + 1)) * t[..., k + 1]
    return a * (b / a) ** t if log else a + (b - a) * t
