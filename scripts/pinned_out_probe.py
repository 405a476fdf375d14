#!/usr/bin/env python3
"""Host-buffer log-L call for mid-size batches: results written by the kernels into mapped pinned host memory (default)
against download commands (RVLL_NO_PINNED_OUT=1).  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evidence_amd import GpuRVModel
from evidence_amd.synthetic import make_workload

w = make_workload(3)
with GpuRVModel(w.fixedpardict, w.table, w.parnames) as m:
    for n in (1024, 4096, 8192, 16384, 32768, 65536):
        theta = w.sample_theta(n, 1)
        ref = None
        for rnd in range(2):
            for mode in ("download", "pinned"):
                if mode == "download":
                    os.environ["RVLL_NO_PINNED_OUT"] = "1"
                else:
                    os.environ.pop("RVLL_NO_PINNED_OUT", None)
                for _ in range(10):
                    out = m.log_likelihood_batch(theta)
                reps = 100
                t0 = time.perf_counter()
                for _ in range(reps):
                    out = m.log_likelihood_batch(theta)
                dt = (time.perf_counter() - t0) / reps
                if ref is None:
                    ref = out
                print(f"n={n:7d} {mode:9s} {dt*1e6:8.1f} us  {n/dt:.3e} evals/s  identical={np.array_equal(out, ref)}", flush=True)
