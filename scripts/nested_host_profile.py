#!/usr/bin/env python3
"""cProfile of bench.py's end-to-end nested-sampling configuration: where the HOST time goes next to the walk call."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evidence_amd import GpuRVModel
from evidence_amd.callbacks import make_ultranest_callbacks, wrapped_params
from evidence_amd.nested import run_nested_slice
from evidence_amd.synthetic import make_workload

w = make_workload(3)
with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
    prior, loglike = make_ultranest_callbacks(m, vectorized=True)
    wr = wrapped_params(m.parnames)
    kw = dict(nlive=32768, kbatch=16384, dlogz=1e-9, max_calls=60_000_000, wrapped=wr, seed=1,
              prior_loglike=m.prior_loglike_batch, walker=m.slice_walk)
    run_nested_slice(prior, loglike, m.ndim, **dict(kw, max_calls=5_000_000))      # warm-up
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    res = run_nested_slice(prior, loglike, m.ndim, **kw)
    pr.disable()
    dt = time.perf_counter() - t0
    print(f"{res.ncall} calls in {dt:.3f} s = {res.ncall / dt:.3e}/s")
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
    # the same with the live set resident on the device (rvll_live_*): what is left on the host
    live_kw = {k: v for k, v in kw.items() if k not in ("prior_loglike", "walker")}
    run_nested_slice(None, None, m.ndim, live=m, **dict(live_kw, max_calls=5_000_000))
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    res = run_nested_slice(None, None, m.ndim, live=m, **live_kw)
    pr.disable()
    dt = time.perf_counter() - t0
    print(f"resident live set: {res.ncall} calls in {dt:.3f} s = {res.ncall / dt:.3e}/s")
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
