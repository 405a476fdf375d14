#!/usr/bin/env python3
"""How much one launch of the cfg3 log-L kernel depends on WHICH 16384 points it gets (run on the GPU box): kernel time
per prior sample (seed) next to the sample's longest Kepler solves (oracle iteration counts of its highest-e points).
A solve at the eccentricity clamp wanders for 30-350 Newton steps: one such point keeps one wave busy for
tens of microseconds, and where it sits in its tile decides how much of that shows as the launch's tail."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evidence_amd import GpuRVModel
from evidence_amd.synthetic import make_workload
from oracle.oracle import OracleModel

w = make_workload(3); B = 16384
ecc_cols = [i for i, n in enumerate(w.parnames) if n.endswith("_ecc")]
with GpuRVModel(w.fixedpardict, w.table, w.parnames) as m:
    orc = OracleModel(m.layout, w.table)
    print("seed   kernel us   points with e >= 0.97   longest solve (steps)   row of it in its 64-point tile")
    for seed in (1234, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11):
        theta = w.sample_theta(B, seed=seed)
        m.dev_upload_theta(theta)
        tm = m.dev_time_loglike(B, warmup=20, iters=200)
        e = theta[:, ecc_cols].max(axis=1)
        rows = np.flatnonzero(e >= 0.97)
        worst, where = 0, -1
        for r in rows:
            it = int(orc.iteration_counts(theta[r]).max())
            if it > worst:
                worst, where = it, int(r % 64)
        print(f"{seed:5d}   {tm['kernel_ms_median'] * 1e3:8.2f}   {rows.size:10d}   {worst:20d}   {where:10d}", flush=True)
