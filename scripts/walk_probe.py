#!/usr/bin/env python3
"""Device-resident slice-sampling walk (rvll_slice_walk): invariants, and end-to-end nested-sampling throughput
with the walk on the GPU vs the host-driven batched walk.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evidence_amd import GpuRVModel
from evidence_amd.callbacks import make_ultranest_callbacks, wrapped_params
from evidence_amd.nested import run_nested_slice
from evidence_amd.synthetic import make_workload

w = make_workload(3)
with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
    rng = np.random.default_rng(0)
    K = 4096
    cube = rng.random((K, m.ndim))
    theta, logl = m.prior_loglike_batch(cube)
    lstar = np.median(logl)
    keep = logl > lstar
    cube, theta, logl = cube[keep], theta[keep], logl[keep]
    d0 = cube - cube.mean(axis=0)
    chol = np.linalg.cholesky(d0.T @ d0 / (len(cube) - 1) + 1e-14 * np.eye(m.ndim))
    wr = wrapped_params(m.parnames)
    t0 = time.perf_counter()
    c2, t2, l2, n = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=3 * m.ndim, seed=7)
    dt = time.perf_counter() - t0
    th_chk, ll_chk = m.prior_loglike_batch(c2)
    print(f"walk: {len(cube)} walkers x {3 * m.ndim} steps: {n} calls in {dt*1e3:.1f} ms = {n/dt:.3e} calls/s; "
          f"all above lstar {bool((l2 > lstar).all())}; theta/logl consistent with end cubes "
          f"{bool(np.array_equal(th_chk, t2))} {bool(np.array_equal(ll_chk, l2))}; moved {float(np.mean(np.any(c2 != cube, axis=1))):.3f}; "
          f"in cube {bool(((c2 >= 0) & (c2 < 1)).all())}")
    c3, t3, l3, n3 = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=3 * m.ndim, seed=7)
    print("deterministic for a seed:", bool(np.array_equal(c2, c3)) and n == n3)

    prior, loglike = make_ultranest_callbacks(m, vectorized=True)
    for nlive, kbatch, max_calls in ((400, 100, 2_000_000), (4096, 1024, 20_000_000), (16384, 4096, 60_000_000),
                                     (16384, 8192, 60_000_000), (65536, 32768, 200_000_000)):
        for name, kw in (("host walk", {}), ("device walk", {"walker": m.slice_walk})):
            if name == "host walk" and nlive > 16384:
                continue
            t0 = time.perf_counter()
            res = run_nested_slice(prior, loglike, m.ndim, nlive=nlive, kbatch=kbatch, dlogz=1e-9, max_calls=max_calls,
                                   wrapped=wr, seed=1, prior_loglike=m.prior_loglike_batch, **kw)
            dt = time.perf_counter() - t0
            print(f"nlive={nlive:6d} kbatch={kbatch:6d} {name:11s}: {res.ncall} calls in {dt:.2f} s = {res.ncall / dt:.3e} calls/s "
                  f"({res.niter} iterations, ln Z so far {res.logz:.2f})", flush=True)

    # where the time goes at 16384 live points: inside the walk call (upload + kernel + download) vs host logic
    spent = {"walk": 0.0, "calls": 0}
    def timed_walk(*a):
        t1 = time.perf_counter()
        out = m.slice_walk(*a)
        spent["walk"] += time.perf_counter() - t1
        spent["calls"] += out[3]
        return out
    t0 = time.perf_counter()
    res = run_nested_slice(prior, loglike, m.ndim, nlive=16384, dlogz=1e-9, max_calls=60_000_000, wrapped=wr, seed=1,
                           prior_loglike=m.prior_loglike_batch, walker=timed_walk)
    dt = time.perf_counter() - t0
    print(f"breakdown nlive=16384: total {dt:.2f} s, inside slice_walk {spent['walk']:.2f} s "
          f"({spent['calls'] / spent['walk']:.3e} calls/s there), host logic {dt - spent['walk']:.2f} s")
