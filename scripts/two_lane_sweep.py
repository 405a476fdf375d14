#!/usr/bin/env python3
"""Two-lane pipelined throughput (launches alternate between the two lanes) by points-per-workgroup."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evidence_amd import GpuRVModel
from evidence_amd.synthetic import CONFIGS, make_workload

for cfg in (3, 5):
    w = make_workload(cfg)
    B = CONFIGS[cfg]["batch"] // (8 if cfg in (4, 5) else 1)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames) as m:
        m.dev_upload_theta(w.sample_theta(B, seed=1))
        for pb in (0, 2, 4, 8, 16, 32):
            if pb * w.table.n_epochs > 16384 and pb > 1:
                continue
            m.set_points_per_block(pb)
            K = 1000 if cfg == 3 else 100
            for lanes in (1, 2):
                for _ in range(K // 5):
                    m.dev_loglike(B)
                    if lanes == 2: m.dev_flip_lane()
                m.dev_sync()
                t0 = time.perf_counter()
                for _ in range(K):
                    m.dev_loglike(B)
                    if lanes == 2: m.dev_flip_lane()
                m.dev_sync()
                dt = (time.perf_counter() - t0) / K
                if m.dev_flip_lane() != 0: m.dev_flip_lane()
                print(f"cfg{cfg} pb={pb:2d} lanes={lanes}: {dt*1e6:8.2f} us/step {B/dt:.3e} evals/s", flush=True)
