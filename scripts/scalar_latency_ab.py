#!/usr/bin/env python3
"""Scalar-callback latency (launch per call, persistent kernel, PolyChord's prior + loglike pair) — for A/B of tile changes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evidence_amd import GpuRVModel
from evidence_amd.callbacks import make_polychord_callbacks
from evidence_amd.synthetic import make_workload
w = make_workload(3)
with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
    x0 = w.sample_theta(4, seed=1)[0]
    out = []
    for mode in ("launch", "server"):
        m.scalar_server(mode == "server")
        for _ in range(300):
            m.log_likelihood(x0)
        t1 = time.perf_counter()
        for _ in range(3000):
            m.log_likelihood(x0)
        out.append(f"{mode} {(time.perf_counter() - t1) / 3000 * 1e6:.2f} us")
    cubes = w.sample_cube(512, seed=3)
    m.scalar_server(True)
    prior, loglike, _, _ = make_polychord_callbacks(m, low_latency=True)
    for c in cubes[:64]:
        loglike(prior(c))
    t1 = time.perf_counter()
    for c in cubes:
        loglike(prior(c))
    out.append(f"pair {(time.perf_counter() - t1) / len(cubes) * 1e6:.2f} us")
    m.scalar_server(False)
    print((sys.argv[1] if len(sys.argv) > 1 else "") + ": " + " | ".join(out))
