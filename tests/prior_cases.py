"""Which golden prior vectors are comparable, and how tightly (shared by CPU and GPU prior tests).

Measured on the MI355X against the golden grid (74 q per prior, incl. 0, 1e-300, 1e-15 ... 1e-9, 1 - 1e-7 ...
1 - 1e-15, 1; profiles/r02_prior_parity.txt): Normal and Alpha bit-identical to scipy (the device evaluates the
Cephes ndtri scipy uses, operation by operation), LogNormal 2e-16, Beta / Gamma <= 3.2e-14 at every point but the
one below, ModJeffreys within one ulp of its power."""
import numpy as np

from evidence_amd import priors as P

# The ONLY excluded point for the special-function kinds: q = 1e-300, where the reference's own library is not
# self-consistent — scipy's beta.ppf(1e-300, 2, 5) = 4.1e-51, although I_x(2, 5) at that x is ~1e-100, not 1e-300
# (the value is 2.6e-151, which is what the device returns), and beta.ppf(1e-300, 0.5, 0.5) returns exactly 0.0 for
# a value of 2.5e-600 that underflows (the device returns the smallest normal number).
Q_EXCLUDED = 1e-300
# north_star bar for Alpha (its formula cancels as q -> 1, evidence/priors.py:375-376); measured: 0 on the grid
TOL = {"Alpha": 1e-10, "Gamma": 2e-13, "Beta": 2e-13}
DEFAULT_TOL = 1e-13


def base_tol(name, args):
    """Per-family bar; a Beta prior with a shape parameter below 0.2 (nothing a config would use: the density is a spike at
    an end point) is held to 1e-11 — measured 7e-13 (host build of the solver) / up to 3.6e-12 (device, verified table
    interpolation, q near 1/2 over many random draws) at Beta(0.1, 20) against scipy, 1.2e-13 at every other shape."""
    if name == "Beta" and min(float(args[0]), float(args[1])) < 0.2:
        return 1e-11
    return TOL.get(name, DEFAULT_TOL)


def tolerance(name, args, ref):
    """Relative tolerance per point.  Alpha is 1/(a - ndtri(q Phi(a))) in the reference: what comes out is the
    reciprocal of a difference that vanishes as q -> 1, so one ulp of the reference's OWN intermediate ndtri value —
    all a different libm's log may legitimately move it by: the device's ndtri agrees with scipy's bit for bit on
    99.93 % of arguments and to 8e-16 otherwise — is a relative change of ulp(a) * |ppf| in the result.  The bar is
    therefore 1e-10 plus four such ulps; on the golden grid (q up to 1 - 1e-15) the device is bit-identical."""
    base = base_tol(name, args)
    if name == "Alpha":
        return base + 4.0 * np.spacing(float(args[0])) * np.abs(np.asarray(ref, float))
    return np.full(np.shape(ref), base)


def spec_for(name, args):
    return getattr(P, name)(*args)


def comparable_mask(name, q, raised):
    m = ~raised
    if name in ("Beta", "Gamma"):
        m &= q != Q_EXCLUDED
    return m


def abs_scale(name, args):
    """ModJeffreys is x0 (1 + xmax/x0)^q - x0 (priors.py:82-83): the subtraction cancels as q -> 0, and the device's
    pow differs from libm's in the last bit for ~15 % of the arguments (never by more than one ulp of the power:
    profiles/r02_prior_parity.txt).  The meaningful error unit is therefore x0 * ulp(power) <= x0 * ulp(1 + xmax/x0);
    expressed as a floor for the relative error's denominator: |got - ref| <= tol * scale with
    scale = x0 * ulp(1 + xmax/x0) / DEFAULT_TOL, i.e. a difference of ONE ulp of the largest power passes, two do not."""
    if name != "ModJeffreys":
        return 0.0
    x0, xmax = float(args[0]), float(args[1])
    return x0 * float(np.spacing(1.0 + xmax / x0)) / DEFAULT_TOL


def rel_err(got, ref, scale=0.0):
    got, ref = np.asarray(got, float), np.asarray(ref, float)
    with np.errstate(all="ignore"):
        e = np.abs(got - ref) / np.maximum(np.maximum(np.abs(ref), scale), 1e-300)
    e = np.where(got == ref, 0.0, e)
    e = np.where(np.isnan(got) & np.isnan(ref), 0.0, e)
    # near a zero crossing (Normal(0,1) at q = 0.5, Uniform(-10,10) ...) use an absolute floor
    return np.where(np.abs(ref) < 1e-6, np.minimum(e, np.abs(got - ref) / 1e-6), e)
