#!/usr/bin/env python3
"""End-to-end nested-sampling throughput of the in-repo batched driver on the cfg3 workload (3 planets, 200
epochs, 19 parameters): likelihood calls per second of wall time including all host-side proposal logic, for a
few live-point counts.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evidence_amd import GpuRVModel
from evidence_amd.callbacks import make_ultranest_callbacks, wrapped_params
from evidence_amd.nested import run_nested_slice
from evidence_amd.synthetic import make_workload

w = make_workload(3)
with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
    prior, loglike = make_ultranest_callbacks(m, vectorized=True)
    for nlive, max_calls in ((400, 2_000_000), (4096, 10_000_000), (16384, 30_000_000)):
        t0 = time.perf_counter()
        res = run_nested_slice(prior, loglike, m.ndim, nlive=nlive, dlogz=1e-9, max_calls=max_calls,
                               wrapped=wrapped_params(m.parnames), seed=1, prior_loglike=m.prior_loglike_batch)
        dt = time.perf_counter() - t0
        print(f"nlive={nlive:6d} kbatch={max(1, nlive // 4):5d}: {res.ncall} likelihood calls in {dt:.2f} s = "
              f"{res.ncall / dt:.3e} calls/s  ({res.niter} iterations, ln Z so far {res.logz:.1f})", flush=True)
