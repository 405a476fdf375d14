#!/usr/bin/env python3
"""Host-buffer log-L call (theta upload + kernel + log-L download) for large batches: whole batch in one go
(RVLL_SPLIT=1) vs 2..16 overlapped chunks vs the built-in choice.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evidence_amd import GpuRVModel
from evidence_amd.synthetic import make_workload

w = make_workload(3)
with GpuRVModel(w.fixedpardict, w.table, w.parnames) as m:
    for n in (8192, 16384, 32768, 65536, 131072, 262144):
        theta = w.sample_theta(n, 1)
        ref = None
        for mode in ("1", "2", "4", "8", "16", "default"):
            if mode == "default":
                os.environ.pop("RVLL_SPLIT", None)
            else:
                os.environ["RVLL_SPLIT"] = mode
            for _ in range(5):
                out = m.log_likelihood_batch(theta)
            reps = 50
            t0 = time.perf_counter()
            for _ in range(reps):
                out = m.log_likelihood_batch(theta)
            dt = (time.perf_counter() - t0) / reps
            if ref is None:
                ref = out
            print(f"n={n:7d} {mode:5s} {dt*1e6:8.1f} us  {n/dt:.3e} evals/s  identical={np.array_equal(out, ref)}", flush=True)

# the same for the cube -> theta -> log-L call (cube up, theta and log-L down)
w = make_workload(3)
with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
    for n in (16384, 65536, 262144):
        cube = w.sample_cube(n, 2)
        ref = None
        for mode in ("1", "2", "4", "8", "default"):
            if mode == "default":
                os.environ.pop("RVLL_SPLIT", None)
            else:
                os.environ["RVLL_SPLIT"] = mode
            for _ in range(3):
                th, ll = m.prior_loglike_batch(cube)
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                th, ll = m.prior_loglike_batch(cube)
            dt = (time.perf_counter() - t0) / reps
            if ref is None:
                ref = (th, ll)
            same = bool(np.array_equal(th, ref[0]) and np.array_equal(ll, ref[1]))
            print(f"prior+loglike n={n:7d} {mode:7s} {dt*1e6:8.1f} us  {n/dt:.3e} evals/s  identical={same}", flush=True)
