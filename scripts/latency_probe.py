#!/usr/bin/env python3
"""Wall-clock latency of the host-buffer calls a sampler makes, by batch size (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evidence_amd import GpuRVModel
from evidence_amd.synthetic import make_workload

w = make_workload(3)
with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
    for n in (1, 16, 256, 1024, 4096, 16384, 65536):
        cube = w.sample_cube(n, 1)
        theta = m.prior_transform_batch(cube)
        m.log_likelihood_batch(theta)
        reps = 200 if n <= 4096 else 30
        t0 = time.perf_counter()
        for _ in range(reps):
            m.log_likelihood_batch(theta)
        t_ll = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            m.prior_transform_batch(cube)
        t_pr = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            m.prior_loglike_batch(cube)
        t_pl = (time.perf_counter() - t0) / reps
        print(f"n={n:6d}  loglike {t_ll*1e6:8.1f} us ({n/t_ll:.3e}/s)   prior {t_pr*1e6:8.1f} us   fused prior+loglike {t_pl*1e6:8.1f} us ({n/t_pl:.3e}/s)", flush=True)
