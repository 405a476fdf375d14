"""Which golden prior vectors are comparable, and how tightly (shared by CPU and GPU prior tests)."""
import numpy as np

from evidence_amd import priors as P

# q beyond these bounds is excluded for the special-function kinds: there the reference's own
# library value is not self-consistent (e.g. scipy's betaincinv(12, 1.5, 1.8e-180) = 7.8e-28 although
# I_x(12, 1.5) at that x is 1e-326), or the formula is ill-conditioned in the reference itself
# (alpha.ppf = 1/(a - ndtri(q Phi(a))) cancels catastrophically as q -> 1).
Q_LO, Q_HI = 1e-12, 1 - 1e-12
TOL = {"Alpha": 2e-10, "Gamma": 5e-13, "Beta": 5e-13}
DEFAULT_TOL = 1e-13


def spec_for(name, args):
    return getattr(P, name)(*args)


def comparable_mask(name, q, raised):
    m = ~raised
    if name in ("Beta", "Gamma"):
        m &= ((q >= Q_LO) & (q <= Q_HI)) | (q == 0) | (q == 1)
    if name == "Alpha":
        m &= ((q >= Q_LO) & (q <= 1 - 1e-6)) | (q == 0) | (q == 1)
    return m


def abs_scale(name, args):
    """Where the reference's own formula subtracts two nearly equal numbers the meaningful error is relative
    to the subtrahend, not to the tiny result: ModJeffreys is x0 (1 + xmax/x0)^q - x0 (priors.py:82-83)."""
    return float(args[0]) if name == "ModJeffreys" else 0.0


def rel_err(got, ref, scale=0.0):
    got, ref = np.asarray(got, float), np.asarray(ref, float)
    with np.errstate(all="ignore"):
        e = np.abs(got - ref) / np.maximum(np.maximum(np.abs(ref), scale), 1e-300)
    e = np.where(got == ref, 0.0, e)
    e = np.where(np.isnan(got) & np.isnan(ref), 0.0, e)
    # near a zero crossing (Normal(0,1) at q = 0.5, Uniform(-10,10) ...) use an absolute floor
    return np.where(np.abs(ref) < 1e-6, np.minimum(e, np.abs(got - ref) / 1e-6), e)
