#!/usr/bin/env python3
"""Latency of the scalar callback loglike(theta) (one point per call, as PolyChord calls it): kernel launch +
stream sync per call vs the persistent scalar-call server.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evidence_amd import GpuRVModel
from evidence_amd.synthetic import make_workload

for cfg in (1, 2, 3, 5):
    w = make_workload(cfg)
    theta = w.sample_theta(2000, seed=3)
    with GpuRVModel(w.fixedpardict, w.table, w.parnames) as m:
        ref = m.log_likelihood_batch(theta)
        out = {}
        for mode in ("launch", "server"):
            m.scalar_server(mode == "server")
            for x in theta[:50]:
                m.log_likelihood(x)
            t0 = time.perf_counter()
            got = np.array([m.log_likelihood(x) for x in theta])
            dt = (time.perf_counter() - t0) / len(theta)
            out[mode] = (dt, bool(np.array_equal(got, ref)))
        # raw C-ABI call without the Python wrapper around it
        import ctypes as C
        from evidence_amd import _abi
        lib = m._lib
        o = np.empty(1); f = np.zeros(1, dtype=np.int32)
        raw = {}
        for mode in ("launch", "server"):
            m.scalar_server(mode == "server")
            x = np.ascontiguousarray(theta[0:1])
            for _ in range(50):
                lib.rvll_loglike_batch(m._h, _abi.as_dp(x), 1, _abi.as_dp(o), _abi.as_ip(f))
            t0 = time.perf_counter()
            for _ in range(2000):
                lib.rvll_loglike_batch(m._h, _abi.as_dp(x), 1, _abi.as_dp(o), _abi.as_ip(f))
            raw[mode] = (time.perf_counter() - t0) / 2000
        # bare round trip: a request the server only acknowledges
        m.scalar_server(True)
        os.environ["RVLL_SERVER_NOOP"] = "1"
        for _ in range(50):
            lib.rvll_loglike_batch(m._h, _abi.as_dp(x), 1, _abi.as_dp(o), _abi.as_ip(f))
        t0 = time.perf_counter()
        for _ in range(2000):
            lib.rvll_loglike_batch(m._h, _abi.as_dp(x), 1, _abi.as_dp(o), _abi.as_ip(f))
        noop = (time.perf_counter() - t0) / 2000
        del os.environ["RVLL_SERVER_NOOP"]
        print(f"cfg{cfg}: request/acknowledge round trip without any evaluation {noop*1e6:.1f} us")
        # prior(cube) then loglike(theta) per point, as PolyChord does
        pair = {}
        m.set_priors(w.priordict())
        cubes = w.sample_cube(1000, seed=5)
        for mode in ("launch", "server"):
            m.scalar_server(mode == "server")
            for c in cubes[:20]:
                m.log_likelihood(m.prior_transform(c))
            t0 = time.perf_counter()
            for c in cubes:
                m.log_likelihood(m.prior_transform(c))
            pair[mode] = (time.perf_counter() - t0) / len(cubes)
        m.scalar_server(False)
        print(f"cfg{cfg}: prior(cube) + loglike(theta) per point: launch {pair['launch']*1e6:.1f} us, server {pair['server']*1e6:.1f} us")
        print(f"cfg{cfg} Ne={w.table.n_epochs} Np={len(m.layout.planets)}: log_likelihood(x) launch {out['launch'][0]*1e6:.1f} us, "
              f"server {out['server'][0]*1e6:.1f} us (bit-identical {out['launch'][1]} {out['server'][1]}); "
              f"raw ctypes call launch {raw['launch']*1e6:.1f} us, server {raw['server']*1e6:.1f} us", flush=True)
