"""Device .ppf of the special-function priors against the golden vectors of the reference, point by point (what\ntests/prior_cases.py turns into tolerances), and the device Cephes ndtri against scipy.special.ndtri.\n\n    python scripts/prior_parity_dump.py > profiles/rNN_prior_parity.txt\n"""
import sys, math, numpy as np
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
import golden, prior_cases as pc
from evidence_amd import GpuRVModel, priors as P
from evidence_amd.data import EpochTable
q, sets = golden.prior_sets()
table = EpochTable.from_arrays(["ia"], [50000.0, 50001.0], [1.0, -1.0], [1.0, 1.0], [0, 0])
for name, args, vals, raised in sets:
    if name not in ("Alpha", "Normal", "LogNormal", "Beta", "Gamma", "ModJeffreys"): continue
    with GpuRVModel({}, table, ["ia_offset"], priordict={"ia_offset": pc.spec_for(name, args)}) as m:
        got = m.prior_transform_batch(q[:, None])[:, 0]
    with np.errstate(all="ignore"):
        rel = np.abs(got - vals) / np.abs(vals)
    rel[got == vals] = 0
    bad = [(float(q[i]), float(vals[i]), float(got[i]), float(rel[i])) for i in range(len(q)) if not raised[i] and rel[i] > 1e-13]
    print(name, args, "identical", float(np.mean(got[~raised] == vals[~raised])), "max rel", float(np.nanmax(rel[~raised & np.isfinite(vals) & (vals != 0)])))
    for b in bad[:14]: print("    q=%r ref=%r got=%r rel=%.2e" % b)
x = np.random.default_rng(1).random(1_000_000)
x = np.concatenate([x, 10.0 ** np.random.default_rng(2).uniform(-300, 0, 200000)])
w = None
from evidence_amd.synthetic import make_workload
wk = make_workload(1)
with GpuRVModel(wk.fixedpardict, wk.table, wk.parnames) as m:
    from scipy import special
    d = m.debug_eval(14, x); r = special.ndtri(x)
    print("device ndtri_cephes vs scipy: identical %.6f, max rel %.2e" % (np.mean(d == r), np.max(np.abs(d - r) / np.abs(r))))
    # ModJeffreys: device pow vs libm pow

# ModJeffreys = x0 (1 + xmax/x0)^q - x0 (evidence/priors.py:82-83): the subtraction cancels as q -> 0, so a last-bit
# difference between the device's pow and libm's shows up divided by the (small) result.  Measured here in units
# of one ulp of the POWER (the quantity both libraries round): what the test tolerance has to absorb.
rng = np.random.default_rng(3)
for x0, xmax in ((1.0, 100.0), (0.5, 2000.0)):
    qq = np.concatenate([rng.random(200000), 10.0 ** rng.uniform(-9, 0, 200000)])
    with GpuRVModel({}, table, ["ia_offset"], priordict={"ia_offset": P.ModJeffreys(x0, xmax)}) as m:
        got = m.prior_transform_batch(qq[:, None])[:, 0]
    ref = x0 * (1 + xmax / x0) ** qq - x0
    power = (1 + xmax / x0) ** qq
    ulps = np.abs(got - ref) / (x0 * np.spacing(power))
    with np.errstate(all="ignore"):
        rel = np.abs(got - ref) / np.abs(ref)
    print(f"ModJeffreys({x0}, {xmax}): device vs libm pow, in ulps of the power: max {ulps.max():.2f}, "
          f"differing in {np.mean(got != ref) * 100:.1f} % of 400000 q; result-relative max {np.nanmax(rel):.2e} "
          f"(at q = {qq[np.nanargmax(rel)]:.3e}), relative to x0 max {np.max(np.abs(got - ref)) / x0:.2e}")
