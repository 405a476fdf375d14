#!/usr/bin/env python3
"""Converged evidence runs at cfg3 (dlogz = 0.5, the reference's UltraNest default): ln Z +- err, iterations, calls, seconds —
live set resident on the device against host-managed, a few seeds each.   python scripts/converged_probe.py [nlive kbatch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evidence_amd import GpuRVModel
from evidence_amd.callbacks import make_ultranest_callbacks, wrapped_params
from evidence_amd.nested import run_nested_slice
from evidence_amd.synthetic import make_workload

nlive = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
kbatch = int(sys.argv[2]) if len(sys.argv) > 2 else nlive // 4
w = make_workload(int(os.environ.get('PROBE_CFG', '3')))
with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
    prior, loglike = make_ultranest_callbacks(m, vectorized=True)
    wr = wrapped_params(m.parnames)
    for seed in (1, 2, 3, 4):
        for name, kw in (("resident", dict(live=m)), ("host-managed", dict(walker=m.slice_walk, prior_loglike=m.prior_loglike_batch))):
            t0 = time.perf_counter()
            r = run_nested_slice(prior, loglike, m.ndim, nlive=nlive, kbatch=kbatch, dlogz=0.5, max_calls=4_000_000_000, wrapped=wr, seed=seed, **kw)
            dt = time.perf_counter() - t0
            print(f"nlive {nlive} kbatch {kbatch} seed {seed} {name:13s}: ln Z = {r.logz:.3f} +- {r.logzerr:.3f}  H = {r.information:.1f}  {r.niter} iterations, "
                  f"{r.ncall} calls, {dt:.2f} s = {r.ncall / dt:.3e} calls/s", flush=True)
