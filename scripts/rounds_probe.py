#!/usr/bin/env python3
"""The walk in its ROUNDS form (rvll_rounds.hip) against the single-kernel forms: same bits, and what each costs.
cfg3, 16384 walkers x 57 moves from a half-prior start (bench.py's per-iteration walk) and smaller walks.

    python scripts/rounds_probe.py [K ...]        (run on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evidence_amd import GpuRVModel
from evidence_amd.callbacks import wrapped_params
from evidence_amd.synthetic import make_workload

SWITCHES = ("RVLL_WALK_ROUNDS", "RVLL_ROUNDS_GROUPS", "RVLL_ROUNDS_FREE", "RVLL_ROUNDS_DEPTH", "RVLL_WALK_SPEC", "RVLL_ROUNDS_W", "RVLL_ROUNDS_PB", "RVLL_ROUNDS_MODE", "RVLL_ROUNDS_FORM", "RVLL_ROUNDS_PRIO", "RVLL_ROUNDS_CHAIN")

def setenv(env):
    for k in SWITCHES:
        os.environ.pop(k, None)
    os.environ.update(env)

w = make_workload(int(os.environ.get("PROBE_CFG", "3")))
q = 0.5
sizes = [int(a) for a in sys.argv[1:]] or [16384, 2048, 100]
forms = [("single-kernel", {"RVLL_WALK_ROUNDS": "0"}),
         ("rounds default (G=2)", {}),
         ("rounds G=1", {"RVLL_ROUNDS_GROUPS": "1"}),
         ("rounds G=3", {"RVLL_ROUNDS_GROUPS": "3"}),
         ("rounds G=4", {"RVLL_ROUNDS_GROUPS": "4"}),
         ("rounds spec 1", {"RVLL_WALK_SPEC": "1"}),
         ("rounds spec 4", {"RVLL_WALK_SPEC": "4"}),
         ("rounds spec 16", {"RVLL_WALK_SPEC": "16"}),
         ("rounds depth 2", {"RVLL_ROUNDS_DEPTH": "2"}),
         ("rounds depth 8", {"RVLL_ROUNDS_DEPTH": "8"}),
         ("rounds free/2", {"RVLL_ROUNDS_FREE": "650"}),
         ("rounds free*2", {"RVLL_ROUNDS_FREE": "2600"})]
with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
    wr = wrapped_params(m.parnames)
    for K in sizes:
        rng = np.random.default_rng(0)
        cube = rng.random((int(K / (1 - q)) + 64, m.ndim))
        theta, logl = m.prior_loglike_batch(cube)
        lstar = np.quantile(logl, q)
        keep = np.flatnonzero(logl > lstar)[:K]
        cube, theta, logl = cube[keep], theta[keep], logl[keep]
        d0 = cube - cube.mean(axis=0)
        chol = np.linalg.cholesky(d0.T @ d0 / (len(cube) - 1) + 1e-14 * np.eye(m.ndim))
        ref = None
        for name, env in forms:
            setenv(env)
            best, out = 1e9, None
            for rep in range(3):
                t0 = time.perf_counter()
                out = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=3 * m.ndim, seed=7)
                best = min(best, time.perf_counter() - t0)
            same = "reference" if ref is None else ("SAME BITS" if all(np.array_equal(a, b) for a, b in zip(out[:3], ref[:3])) and out[3] == ref[3] else "DIFFERENT")
            if ref is None:
                ref = out
            print(f"K={K:6d} {name:20s}: {out[3]} calls, {m.slice_walk_evaluated()} slots, {m.slice_walk_rounds():5d} rounds, "
                  f"best of 3 {best * 1e3:8.2f} ms = {out[3] / best:.3e} calls/s  [{same}]", flush=True)
        setenv({})
