#!/usr/bin/env python3
"""cube -> theta -> log-L: one launch (prior transform in the log-L kernel's staging step) vs two launches
(prior kernels, then log-L kernel), device-resident, by batch size.  Run on the GPU box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evidence_amd import GpuRVModel
from evidence_amd.synthetic import make_workload

for cfg in (3, 2):
    w = make_workload(cfg)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        for B in (1, 16, 64, 256, 1024, 4096, 16384, 65536):
            m.dev_fill_cube(B, seed=1)
            res = {}
            for name, step in (("two", lambda: (m.dev_prior(B), m.dev_loglike(B))), ("one", lambda: m.dev_prior_loglike(B))):
                for _ in range(20):
                    step()
                m.dev_sync()
                reps = 200
                t0 = time.perf_counter()
                for _ in range(reps):
                    step()
                m.dev_sync()
                res[name] = (time.perf_counter() - t0) / reps * 1e6
                # latency of a single dependent step (launch -> sync)
                t0 = time.perf_counter()
                for _ in range(50):
                    step(); m.dev_sync()
                res[name + "_sync"] = (time.perf_counter() - t0) / 50 * 1e6
            print(f"cfg{cfg} B={B:6d}  back-to-back us/step: two={res['two']:.1f} one={res['one']:.1f}   "
                  f"launch+sync us: two={res['two_sync']:.1f} one={res['one_sync']:.1f}", flush=True)
