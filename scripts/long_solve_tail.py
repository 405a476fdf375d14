#!/usr/bin/env python3
"""One long Kepler solve is a launch's tail: microseconds per launch of the cfg3 log-L kernel on five draws
of 16384 points — three with a wandering solve (cube99, s24; s1234 mildly) and two without (s9, s21) — wall clock over
1500 back-to-back launches after a time-based warm-up, two passes.  Run on the GPU box; RVLL_FORM=tile|cu selects the form.
    python scripts/long_solve_tail.py [label]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evidence_amd import GpuRVModel, FLAG_WANDERED
from evidence_amd.synthetic import make_workload

w = make_workload(3); B = 16384
label = sys.argv[1] if len(sys.argv) > 1 else "this tree"
with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
    draws = {}
    m.dev_fill_cube(B, 99); m.dev_prior(B); m.dev_sync()
    draws["cube99"] = m.dev_download(B, theta=True)[0]
    for seed in (1234, 24, 9, 21):
        draws[f"s{seed}"] = w.sample_theta(B, seed=seed)
    t0 = time.perf_counter()
    m.dev_upload_theta(draws["s9"])
    while time.perf_counter() - t0 < 1.0:            # clocks
        for _ in range(200):
            m.dev_loglike(B)
        m.dev_sync()
    info = []
    for name in ("s1234", "cube99", "s24", "s9", "s21"):
        _, fl = m.log_likelihood_batch(draws[name], return_flags=True)
        info.append(f"{name} {int(((fl & FLAG_WANDERED) != 0).sum())}")
    print(f"# {label}: points flagged RVLL_FLAG_WANDERED per draw: " + ", ".join(info))
    for _ in range(2):
        row = []
        for name in ("s1234", "cube99", "s24", "s9", "s21"):
            m.dev_upload_theta(draws[name])
            for _ in range(300):
                m.dev_loglike(B)
            m.dev_sync()
            t1 = time.perf_counter()
            for _ in range(1500):
                m.dev_loglike(B)
            m.dev_sync()
            row.append(f"{name} {(time.perf_counter() - t1) / 1500 * 1e6:6.2f}")
        print("#   " + " | ".join(row), flush=True)
