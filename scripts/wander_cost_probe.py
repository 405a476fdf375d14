#!/usr/bin/env python3
"""What the exact redo of wandering Kepler solves (rvll_set_wander_exact, default on) costs a launch, draw by draw: kernel time
with the redo on and off, the number of points flagged RVLL_FLAG_WANDERED, for cfg3 batches drawn through the prior transform
and for the cfg4 / cfg5 shard batches of bench.py.      python scripts/wander_cost_probe.py      (run on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evidence_amd import GpuRVModel, _abi
from evidence_amd.synthetic import make_workload

WANDERED = _abi.FLAG_WANDERED if hasattr(_abi, "FLAG_WANDERED") else 4


def timed(m, n):
    return m.dev_time_loglike(n, warmup=5, iters=30)["kernel_ms_median"] * 1e3


def report(label, m, n):
    m.set_wander_exact(False)
    m.dev_loglike(n); m.dev_sync()
    flags = m.dev_download(n, logl=False, flags=True)[2]
    t_off = timed(m, n)
    m.set_wander_exact(True)
    t_on = timed(m, n)
    print(f"{label:44s}: redo off {t_off:8.1f} us, on {t_on:8.1f} us (+{(t_on / t_off - 1) * 100:5.1f} %), {int(np.count_nonzero(flags & WANDERED)):3d} of {n} points wandered", flush=True)


w = make_workload(3)
with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
    for _ in range(300):                      # clocks
        m.dev_fill_cube(16384, seed=1); m.dev_prior(16384); m.dev_loglike(16384)
    m.dev_sync()
    m.dev_upload_theta(w.sample_theta(16384, seed=1234))
    report("cfg3 16384, bench.py's headline batch", m, 16384)
    for seed in range(1, 13):
        m.dev_fill_cube(16384, seed=seed)
        m.dev_prior(16384)
        report(f"cfg3 16384, prior draw seed {seed}", m, 16384)
for cfg, b in ((4, 8192), (5, 16384)):
    wk = make_workload(cfg)
    with GpuRVModel(wk.fixedpardict, wk.table, wk.parnames) as m:
        for seed in (4321, 2, 3, 4):
            m.dev_upload_theta(wk.sample_theta(b, seed=seed))
            report(f"cfg{cfg} shard {b}, theta seed {seed}", m, b)
