"""Prior specifications: the reference's prior NAMES and arguments, compiled to the
descriptors the device prior-transform kernel consumes.

Same vocabulary as evidence/priors.py:429-467 (`Uniform`, `Jeffreys`, ... ) and the
same config entry format `[value, jump_flag, [PriorName, *args]]` handled by
`prior_constructor` (evidence/priors.py:472-505).  A PriorSpec is inert data: the
transform itself (`.ppf` in the reference) runs on the GPU through
rvll_prior_batch — there is no host implementation of any quantile function here.

Table-inverted priors (evidence/priors.py:118-124, 195-202, 223-228, 282-287,
321-326, 349-354, 138-144) are piecewise-linear inverses of the analytic CDF on a
10^4-step grid that the reference REBUILDS ON EVERY CALL; here the same grid is
built once at construction (`numpy.arange` with the same arguments, `scipy.special.ndtr`
for the normal CDF — the function `scipy.stats.norm.cdf` evaluates) and uploaded.
"""
import math
from dataclasses import dataclass, field
from typing import Optional, Tuple

import numpy as np

from . import _abi

# evidence/priors.py:9-10
_N = 1e4
_STEP = 1.0 / _N


class PriorError(Exception):
    """Unknown prior name or invalid arguments (evidence/priors.py:15-16, 499-501)."""


@dataclass
class PriorSpec:
    name: str
    kind: int
    args: Tuple[float, ...]
    table_cdf: Optional[np.ndarray] = None     # sorted knots (TABLE kinds)
    table_x: Optional[np.ndarray] = None
    table_post: int = 0
    group: int = -1
    _keep: list = field(default_factory=list, repr=False)

    def to_c(self):
        p = _abi.Prior()
        p.kind = self.kind
        p.group = self.group
        for i in range(_abi.PRIOR_NARGS):
            p.args[i] = float(self.args[i]) if i < len(self.args) else 0.0
        if self.table_cdf is not None:
            cdf = np.ascontiguousarray(self.table_cdf, dtype=np.float64)
            x = np.ascontiguousarray(self.table_x, dtype=np.float64)
            self._keep = [cdf, x]
            p.table_cdf = _abi.as_dp(cdf)
            p.table_x = _abi.as_dp(x)
            p.table_n = int(cdf.shape[0])
        p.table_post = int(self.table_post)
        return p


def _f(*vals):
    return tuple(float(v) for v in vals)


def _require(cond, name, msg):
    if not cond:
        raise PriorError(f"{name}: {msg}")


# ---- closed forms ------------------------------------------------------------------
def Uniform(xmin, xmax):                         # evidence/priors.py:22-42
    _require(xmin < xmax, "Uniform", "needs xmin < xmax")
    return PriorSpec("Uniform", _abi.PRIOR_UNIFORM, _f(xmin, xmax))


def Jeffreys(xmin, xmax):                        # :45-63
    _require(xmin > 0 and xmax > xmin, "Jeffreys", "needs 0 < xmin < xmax")
    return PriorSpec("Jeffreys", _abi.PRIOR_JEFFREYS, _f(xmin, xmax))


def ModJeffreys(x0, xmax):                       # :66-83
    _require(xmax > x0 > 0, "ModJeffreys", "needs 0 < x0 < xmax")
    return PriorSpec("ModJeffreys", _abi.PRIOR_MODJEFFREYS, _f(x0, xmax))


def UniformFrequency(xmin, xmax):                # :85-101
    _require(xmax > xmin > 0, "UniformFrequency", "needs 0 < xmin < xmax")
    return PriorSpec("UniformFrequency", _abi.PRIOR_UNIFORMFREQUENCY, _f(xmin, xmax))


def TruncatedRayleigh(sigma, xmax):              # :231-252
    _require(sigma > 0, "TruncatedRayleigh", "needs sigma > 0")
    return PriorSpec("TruncatedRayleigh", _abi.PRIOR_TRUNCRAYLEIGH, _f(sigma, xmax))


# ---- scipy.stats families -----------------------------------------------------------
def Normal(loc=0.0, scale=1.0):                  # :436  (= scipy.stats.norm)
    _require(scale > 0, "Normal", "needs scale > 0")
    return PriorSpec("Normal", _abi.PRIOR_NORMAL, _f(loc, scale))


def LogNormal(s, loc=0.0, scale=1.0):            # :437  (= scipy.stats.lognorm)
    _require(s > 0 and scale > 0, "LogNormal", "needs s > 0 and scale > 0")
    return PriorSpec("LogNormal", _abi.PRIOR_LOGNORMAL, _f(s, loc, scale))


def Beta(a, b):                                  # :378-398 (scipy.stats.beta.ppf)
    _require(a > 0 and b > 0, "Beta", "needs a > 0 and b > 0")
    lbeta = math.lgamma(a) + math.lgamma(b) - math.lgamma(a + b)     # setup-time constant for the kernel
    return PriorSpec("Beta", _abi.PRIOR_BETA, _f(a, b, lbeta))


def Gamma(alpha, beta):                          # :400-425 (scipy.stats.gamma.ppf, scale = 1/beta)
    _require(alpha > 0 and beta > 0, "Gamma", "needs alpha > 0 and beta > 0")
    return PriorSpec("Gamma", _abi.PRIOR_GAMMA, _f(alpha, beta, math.lgamma(alpha)))


def Alpha(a):                                    # :357-376 (scipy.stats.alpha.ppf)
    _require(a > 0, "Alpha", "needs a > 0")
    return PriorSpec("Alpha", _abi.PRIOR_ALPHA, _f(a, 0.5 * math.erfc(-a / math.sqrt(2.0))))   # Phi(a)


# ---- table-inverted families -----------------------------------------------------------
def _ndtr(z):
    from scipy.special import ndtr            # what scipy.stats.norm.cdf evaluates
    return ndtr(z)


def _table(name, x, cdf, lo, hi, wrapped, post=0):
    """Knots of interp1d(cdf, x): sorted by cdf with a stable sort (scipy's interp1d sorts
    with argsort(kind='mergesort') when assume_sorted is False)."""
    x = np.asarray(x, dtype=np.float64)
    cdf = np.asarray(cdf, dtype=np.float64)
    order = np.argsort(cdf, kind="mergesort")
    return PriorSpec(name, _abi.PRIOR_TABLE, _f(lo, hi, 1.0 if wrapped else 0.0),
                     table_cdf=cdf[order], table_x=x[order], table_post=post)


def _grid(xmin, xmax):
    dx = (xmax - xmin) * _STEP
    return np.arange(xmin, xmax + dx, dx)


def Binormal(mu1, sigma1, mu2, sigma2, A):       # :103-124
    _require(sigma1 > 0 and sigma2 > 0 and mu1 <= mu2 and -1.0 <= A <= 1.0, "Binormal",
             "needs sigma1, sigma2 > 0, mu1 <= mu2, |A| <= 1")
    x = _grid(mu1 - 9.0 * sigma1, mu2 + 9.0 * sigma2)
    n1 = _ndtr((x - mu1) / sigma1)
    n2 = _ndtr((x - mu2) / sigma2)
    cdf = 0.5 * (n1 * (1.0 - A) + n2 * (1.0 + A))
    return _table("Binormal", x, cdf, -np.inf, np.inf, wrapped=True)


def AsymmetricNormal(mu, sigma1, sigma2):        # :176-202
    _require(sigma1 > 0 and sigma2 > 0, "AsymmetricNormal", "needs sigma1, sigma2 > 0")
    x = _grid(mu - 9 * sigma1, mu + 9 * sigma2)
    k1 = 2.0 * sigma1 / (sigma1 + sigma2)
    k2 = 2.0 * sigma2 / (sigma1 + sigma2)
    left = _ndtr((x - mu) / sigma1) * k1
    right = (_ndtr((x - mu) / sigma2) - 0.5) * k2
    cdf = np.where(x <= mu, left, k1 * 0.5 + right)
    return _table("AsymmetricNormal", x, cdf, -np.inf, np.inf, wrapped=True)


def TruncatedUNormal(mu, sigma, xmin, xmax):     # :205-228 (overrides ppf: no scipy front end)
    _require(sigma > 0, "TruncatedUNormal", "needs sigma > 0")
    x = _grid(xmin, xmax)
    lo = _ndtr((xmin - mu) / sigma)
    norm = _ndtr((xmax - mu) / sigma) - lo
    cdf = (_ndtr((x - mu) / sigma) - lo) / norm
    cdf = np.where(x >= xmin, cdf, 0.0)
    cdf = np.where(x < xmax, cdf, 1.0)
    return _table("TruncatedUNormal", x, cdf, np.nan, np.nan, wrapped=False)


def PowerLaw(alpha, xmin, xmax):                 # :266-287
    _require(xmax > xmin and xmin >= 0 and xmax > 0 and alpha != -1, "PowerLaw",
             "needs 0 <= xmin < xmax and alpha != -1")
    x = _grid(xmin, xmax)
    scale = 1.0 / (xmax ** (1.0 + alpha) - xmin ** (1.0 + alpha))
    cdf = scale * (x ** (1.0 + alpha) - xmin ** (1.0 + alpha))
    cdf = np.where(x > xmin, cdf, 0.0)
    cdf = np.where(x >= xmax, 1.0, cdf)
    return _table("PowerLaw", x, cdf, -np.inf, np.inf, wrapped=True)


def DoublePowerLaw(alpha, beta, x0, xmin, xmax):  # :290-326
    _require(xmax > xmin and xmin >= 0 and xmax > 0 and alpha != -1, "DoublePowerLaw",
             "needs 0 <= xmin < xmax and alpha != -1")
    x = _grid(xmin, xmax)
    a1 = (x0 ** (1.0 + alpha) - xmin ** (1.0 + alpha)) / (alpha + 1.0)
    a2 = (xmax ** (1.0 + beta) - x0 ** (1.0 + beta)) / (beta + 1.0)
    join = (x0 * 1.0) ** alpha / (x0 * 1.0) ** beta
    A = 1.0 / (a1 + join * a2)
    low = A * (x ** (1.0 + alpha) - xmin ** (1.0 + alpha)) / (1.0 + alpha)
    high = (A * (x0 ** (1.0 + alpha) - xmin ** (1.0 + alpha)) / (1.0 + alpha)
            + join * A * (x ** (1.0 + beta) - x0 ** (1.0 + beta)) / (1.0 + beta))
    cdf = np.where(x < x0, low, high)
    cdf = np.where(x > xmin, cdf, 0.0)
    cdf = np.where(x >= xmax, 1.0, cdf)
    return _table("DoublePowerLaw", x, cdf, -np.inf, np.inf, wrapped=True)


def Sine(xmin, xmax):                            # :329-354 (support a=0, b=180 degrees, :456-457)
    _require(xmax > xmin, "Sine", "needs xmin < xmax")
    x = _grid(xmin, xmax)
    lo = max(float(xmin), 0.0)
    hi = min(float(xmax), 180.0)
    rad = np.pi / 180.0
    span = np.cos(lo * rad) - np.cos(hi * rad)
    cdf = (np.cos(lo * rad) - np.cos(x * rad)) / span
    cdf = np.where(x >= lo, cdf, 0.0)
    cdf = np.where(x <= hi, cdf, 1.0)
    return _table("Sine", x, cdf, 0.0, 180.0, wrapped=True)


def Log10Normal(mu, sigma):                      # :127-144
    """Upstream `_ppf` calls numpy.linspace with a float count (N = 1e4) and raises
    TypeError on any current numpy; this builds the evident intent, linspace(.., int(N)).
    Parity: n/a (the reference raises)."""
    _require(sigma > 0, "Log10Normal", "needs sigma > 0")
    x = np.linspace(mu - 9.0 * sigma, mu + 9.0 * sigma, int(_N))
    cdf = _ndtr((np.log10(10 ** x) - mu) / sigma)
    return _table("Log10Normal", x, cdf, -np.inf, np.inf, wrapped=True, post=1)


# ---- forced-identifiability (sorted) priors -----------------------------------------------
def SortedUniform(a, b):                         # :462-466 (pypolychord.priors.SortedUniformPrior)
    _require(a < b, "SortedUniform", "needs a < b")
    return PriorSpec("SortedUniform", _abi.PRIOR_SORTED_UNIFORM, _f(a, b), group=0)


def SortedLogUniform(a, b):                      # :462-467 (pypolychord.priors.LogSortedUniformPrior)
    _require(0 < a < b, "SortedLogUniform", "needs 0 < a < b")
    return PriorSpec("SortedLogUniform", _abi.PRIOR_SORTED_LOGUNIFORM, _f(a, b), group=1)


# name -> factory, as evidence/priors.py:470 (`distdict = globals().copy()`)
distdict = {
    "Uniform": Uniform, "Jeffreys": Jeffreys, "ModJeffreys": ModJeffreys,
    "UniformFrequency": UniformFrequency, "Normal": Normal, "LogNormal": LogNormal,
    "Log10Normal": Log10Normal, "Binormal": Binormal, "AsymmetricNormal": AsymmetricNormal,
    "TruncatedUNormal": TruncatedUNormal, "TruncatedRayleigh": TruncatedRayleigh,
    "PowerLaw": PowerLaw, "DoublePowerLaw": DoublePowerLaw, "Sine": Sine, "Alpha": Alpha,
    "Beta": Beta, "Gamma": Gamma, "SortedUniform": SortedUniform, "SortedLogUniform": SortedLogUniform,
}


def prior_constructor(input_dict, customprior_dict=None):
    """{object: {par: [value, flag, [PriorName, *args]]}} -> {"object_par": PriorSpec}.

    Same walk as evidence/priors.py:472-505: non-list entries and flag-0 (fixed)
    parameters are skipped; an unknown prior name raises PriorError.
    """
    priordict = {}
    for objkey, pars in input_dict.items():
        for parkey, parlist in pars.items():
            if not isinstance(parlist, list):
                continue
            if parlist[1] == 0:
                continue
            priortype, args = parlist[2][0], parlist[2][1:]
            try:
                factory = distdict[priortype]
            except KeyError:
                raise PriorError(f"Parameter {objkey}_{parkey}: Unknown type of prior.") from None
            priordict[f"{objkey}_{parkey}"] = factory(*args)
    return priordict
