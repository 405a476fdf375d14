#!/usr/bin/env python3
"""Kernel time of the fused log-L launch for every BASELINE config and a sweep of points-per-workgroup
(run on the GPU box).  Prints one line per (config, pb): HIP-event mean kernel ms and evals/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evidence_amd import GpuRVModel
from evidence_amd.synthetic import CONFIGS, make_workload

for cfg in (1, 2, 3, 4, 5):
    w = make_workload(cfg)
    B = CONFIGS[cfg]["batch"] // (8 if cfg in (4, 5) else 1)
    theta = w.sample_theta(B, seed=1)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames) as m:
        m.dev_upload_theta(theta)
        m.dev_time_loglike(B, warmup=10, iters=30)          # warm clocks / caches before the sweep
        for pb in (0, 1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32):
            if pb * w.table.n_epochs > 16384 and pb > 1:
                continue
            m.set_points_per_block(pb)
            t = m.dev_time_loglike(B, warmup=3, iters=30)
            print(f"cfg{cfg} B={B:6d} Ne={w.table.n_epochs:5d} pb={pb:2d}->{t['points_per_block']:2d} blocks={t['blocks']:6d} "
                  f"kernel_ms={t['kernel_ms_mean']:.4f} evals/s={B / t['kernel_ms_mean'] * 1e3:.3e}", flush=True)
