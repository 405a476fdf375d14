#!/usr/bin/env python3
"""cube -> theta -> log-L in one launch (the fused-slim kernels) at cfg3, 16384 and 2048 points, exact redo off: for A/B of kernel builds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evidence_amd import GpuRVModel
from evidence_amd.synthetic import make_workload
w = make_workload(3)
with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
    m.set_wander_exact(False)
    out = []
    for n, reps in ((16384, 300), (2048, 1000)):
        m.dev_fill_cube(n, seed=5)
        for _ in range(200):
            m.dev_prior_loglike(n)
        m.dev_sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            m.dev_prior_loglike(n)
        m.dev_sync()
        out.append(f"{n}: {(time.perf_counter() - t0) / reps * 1e6:.2f} us")
    print((sys.argv[1] if len(sys.argv) > 1 else "") + " " + " | ".join(out))
