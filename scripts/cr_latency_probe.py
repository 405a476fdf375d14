#!/usr/bin/env python3
"""Latency of the correctly rounded sin / cos pair on the device: chains of 1000 dependent evaluations in ONE wave (debug_eval ops
15 - 19), the situation of a wandering Kepler solve in the redo pass (one lane of one wave busy, nothing to hide latency behind).
    python scripts/cr_latency_probe.py      (run on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evidence_amd import GpuRVModel
from evidence_amd.synthetic import make_workload

w = make_workload(3)
with GpuRVModel(w.fixedpardict, w.table, w.parnames) as m:
    for lanes in (1, 64):
        for op, name, x0 in ((15, "sincos_cr, |x| ~ 1e12", 1.234e12), (15, "sincos_cr, |x| ~ 3", 3.1), (16, "reduce_dd alone, |x| ~ 1e12", 1.234e12),
                             (17, "table kernel alone", 0.4), (18, "series kernel alone", 0.4), (19, "sincos_any, |x| ~ 1e12", 1.234e12), (19, "sincos_any, |x| ~ 3", 3.1)):
            x = np.full(lanes, x0) * (1 + 1e-3 * np.arange(lanes))
            m.debug_eval(op, x)
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                m.debug_eval(op, x)
                best = min(best, time.perf_counter() - t0)
            base = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                m.debug_eval(0, x)
                base = min(base, time.perf_counter() - t0)
            print(f"{lanes:2d} lanes, {name:30s}: {(best - base) * 1e6 / 1000:7.3f} us per evaluation", flush=True)
