#!/usr/bin/env python3
"""Device against the oracle where the reference's Newton iteration is least forgiving: eccentricities 0.95 .. 0.9925 of
one planet of the golden sweep's model (tests/golden/loglike_high_ecc.npz), thousands of random points; beside it, how
far the oracle itself moves when its sin / cos are nudged by one unit in the last place.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import golden
from evidence_amd import GpuRVModel
from oracle import oracle as orc

case = golden.high_ecc_case()
names = case.parnames
rng = np.random.default_rng(99)
n = 20000
lo, hi = case.theta.min(axis=0), case.theta.max(axis=0)
theta = rng.uniform(lo, hi, (n, len(names)))
ie = names.index("planet1_ecc")
theta[:, ie] = rng.uniform(0.95, 0.9925, n)
theta[:, names.index("planet2_ecc")] = rng.beta(0.867, 3.03, n)
with GpuRVModel(case.fixed, case.table, names) as m:
    got = m.log_likelihood_batch(theta)
om = orc.OracleModel(case.layout, case.table)
ref = om.loglike(theta, nthreads=8)
cond = om.conditioning(theta, nthreads=8, eps=-2.0 ** -53)
err = golden.rel_err(got, ref)
for a, b in ((0.95, 0.97), (0.97, 0.98), (0.98, 0.985), (0.985, 0.99), (0.99, 0.9926)):
    sel = (theta[:, ie] >= a) & (theta[:, ie] < b)
    print(f"e in [{a}, {b}): {sel.sum():5d} points; device vs oracle: max {err[sel].max():.2e}, > 1e-10: {int((err[sel] > 1e-10).sum())}, "
          f"> 1e-9: {int((err[sel] > 1e-9).sum())};  oracle vs oracle with sin/cos one ulp down: max {cond[sel].max():.2e}, "
          f"> 1e-10: {int((cond[sel] > 1e-10).sum())}")
