#!/usr/bin/env python3
"""One large rvll_slice_walk call at cfg3 and bench.py's end-to-end nested-sampling configuration, timed: calls/s inside
the walk call, candidates per move, and how many of the evaluated tile slots were used (the rest were candidates
evaluated ahead for a rejection that did not come).  Run on the GPU box.

    python scripts/walk_phase_probe.py [--walkers 16384] [--nsteps 57]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evidence_amd import GpuRVModel
from evidence_amd.callbacks import wrapped_params
from evidence_amd.synthetic import make_workload

ap = argparse.ArgumentParser()
ap.add_argument("--walkers", type=int, default=16384)
ap.add_argument("--nsteps", type=int, default=57)
ap.add_argument("--quantile", type=float, default=0.5)
ap.add_argument("--ahead", type=int, default=0, help="rvll_set_walk_speculation (0: library default)")
ap.add_argument("--pb", type=int, default=0, help="walkers per workgroup (0: library default)")
args = ap.parse_args()
w = make_workload(3)
with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
    if args.ahead:
        m.set_walk_speculation(args.ahead)
    if args.pb:
        m.set_points_per_block(args.pb)
    rng = np.random.default_rng(0)
    K = args.walkers
    cube = rng.random((int(K / (1 - args.quantile)) + 64, m.ndim))
    theta, logl = m.prior_loglike_batch(cube)
    lstar = np.quantile(logl, args.quantile)
    keep = np.flatnonzero(logl > lstar)[:K]
    cube, theta, logl = cube[keep], theta[keep], logl[keep]
    d0 = cube - cube.mean(axis=0)
    chol = np.linalg.cholesky(d0.T @ d0 / (len(cube) - 1) + 1e-14 * np.eye(m.ndim))
    wr = wrapped_params(m.parnames)
    for rep in range(3):
        t0 = time.perf_counter()
        c2, t2, l2, n = m.slice_walk(cube, theta, logl, lstar, chol, wr, nsteps=args.nsteps, seed=7)
        dt = time.perf_counter() - t0
        ev = m.slice_walk_evaluated()
        extra = f"; tile slots evaluated {ev} ({n / ev:.3f} of them used)"
        print(f"{len(cube)} walkers x {args.nsteps} moves: {n} calls ({n / len(cube) / args.nsteps:.2f} per move) in {dt*1e3:.1f} ms "
              f"= {n/dt:.3e} calls/s{extra}", flush=True)
        ph = m.slice_walk_phases()
        if ph[4]:
            names = ("directions + chord limits", "candidates", "prior transform + log-L tile", "accept / copy / bookkeeping")
            tot_t = sum(ph[:4])
            print("    phase clock (diagnostic build): workgroup life mean %.2f ms, longest %.2f ms; " % (tot_t / ph[4] * 1e-5, ph[5] * 1e-5) +
                  ", ".join(f"{nm} {100 * v / tot_t:.1f} %" for nm, v in zip(names, ph[:4])), flush=True)

    # the same bookkeeping over a nested-sampling run (bench.py's end-to-end configuration)
    from evidence_amd.callbacks import make_ultranest_callbacks
    from evidence_amd.nested import run_nested_slice
    prior, loglike = make_ultranest_callbacks(m, vectorized=True)
    tot = {"calls": 0, "slots": 0, "t": 0.0, "moves": 0}
    def walker(*a):
        t1 = time.perf_counter()
        out = m.slice_walk(*a)
        tot["t"] += time.perf_counter() - t1
        tot["slots"] += m.slice_walk_evaluated()
        tot["calls"] += int(out[3])
        tot["moves"] += len(a[0]) * a[6]
        return out
    t0 = time.perf_counter()
    res = run_nested_slice(prior, loglike, m.ndim, nlive=32768, kbatch=16384, dlogz=1e-9, max_calls=60_000_000, wrapped=wr,
                           seed=1, prior_loglike=m.prior_loglike_batch, walker=walker)
    dt = time.perf_counter() - t0
    fill = f", {tot['calls'] / tot['slots']:.3f} of the evaluated tile slots used"
    print(f"nested 32768/16384: {res.ncall} calls in {dt:.2f} s = {res.ncall / dt:.3e}/s; inside the walk {tot['calls'] / tot['t']:.3e}/s, "
          f"{tot['calls'] / tot['moves']:.2f} candidates per move{fill}")
