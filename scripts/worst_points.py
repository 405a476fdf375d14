#!/usr/bin/env python3
"""Diagnose the largest GPU-vs-oracle log-L differences in a big seeded batch (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evidence_amd import GpuRVModel
from evidence_amd.synthetic import make_workload
from oracle.oracle import OracleModel

w = make_workload(3)
n = 400_000
theta = w.sample_theta(n, seed=2024)
with GpuRVModel(w.fixedpardict, w.table, w.parnames) as m:
    got = m.log_likelihood_batch(theta)
    layout = m.layout
om = OracleModel(layout, w.table)
ref = om.loglike(theta, nthreads=16)
err = np.abs(got - ref) / np.abs(ref)
order = np.argsort(err)[::-1][:12]
ie = [w.parnames.index(f"planet{k}_ecc") for k in (1, 2, 3)]
ik = [w.parnames.index(f"planet{k}_k1") for k in (1, 2, 3)]
ip = [w.parnames.index(f"planet{k}_period") for k in (1, 2, 3)]
print("percentiles of rel err: 50%% %.2e 99%% %.2e 99.99%% %.2e max %.2e" % tuple(np.percentile(err, [50, 99, 99.99, 100])))
for i in order:
    it = om.iteration_counts(theta[i])
    print(f"pt {i:6d} rel {err[i]:.2e} abs {abs(got[i]-ref[i]):.2e} logL {ref[i]:.4e}  ecc {theta[i, ie].round(4)}  K {theta[i, ik].round(2)}  "
          f"P {theta[i, ip].round(3)}  max steps/planet {it.max(axis=1)}")
