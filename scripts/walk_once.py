#!/usr/bin/env python3
"""A few large rvll_slice_walk calls at cfg3 (16384 walkers x 57 moves from a half-prior start) and nothing else: the
program rocprofv3 counts the walk kernels of (scripts/_walk_pmc.sh).  RVLL_WALK_ROWS selects the form of the second part."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evidence_amd import GpuRVModel
from evidence_amd.callbacks import wrapped_params
from evidence_amd.synthetic import make_workload

w = make_workload(3)
q = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
    rng = np.random.default_rng(0)
    K = 16384
    cube = rng.random((int(K / (1 - q)) + 64, m.ndim))
    theta, logl = m.prior_loglike_batch(cube)
    lstar = np.quantile(logl, q)
    keep = np.flatnonzero(logl > lstar)[:K]
    cube, theta, logl = cube[keep], theta[keep], logl[keep]
    d0 = cube - cube.mean(axis=0)
    chol = np.linalg.cholesky(d0.T @ d0 / (len(cube) - 1) + 1e-14 * np.eye(m.ndim))
    wr = wrapped_params(m.parnames)
    for rep in range(4):
        t0 = time.perf_counter()
        c2, t2, l2, n = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=57, seed=7)
        dt = time.perf_counter() - t0
        print(f"{K} walkers x 57 moves at quantile {q}: {n} calls in {dt*1e3:.1f} ms = {n/dt:.3e}/s, {m.slice_walk_evaluated()} slots", flush=True)
