#!/usr/bin/env python3
"""FIP periodogram accumulation (fip_criterion.py:305-339): GPU kernels vs the C fold vs the reference-style
numpy loop, on a synthetic set of posteriors of realistic size.  Run on the GPU box.

    python3 scripts/fip_bench.py [--runs 5] [--nmod 4] [--samples 50000] [--tobs 1000]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evidence_amd import fip

ap = argparse.ArgumentParser()
ap.add_argument("--runs", type=int, default=5)
ap.add_argument("--nmod", type=int, default=4)
ap.add_argument("--samples", type=int, default=50000)
ap.add_argument("--tobs", type=float, default=1000.0)
ap.add_argument("--pmin", type=float, default=1.5)
ap.add_argument("--pmax", type=float, default=1000.0)
ap.add_argument("--repeats", type=int, default=20)
ap.add_argument("--no-cpu", action="store_true")
ap.add_argument("--peak-frac", type=float, default=0.8, help="fraction of the samples sitting on the posterior peaks")
args = ap.parse_args()

rng = np.random.default_rng(2021)
peaks = np.exp(rng.uniform(np.log(args.pmin * 2), np.log(args.pmax / 2), args.nmod))
post = []
for r in range(args.runs):
    per_k = [None]
    for k in range(1, args.nmod):
        n = args.samples
        s = np.exp(rng.uniform(np.log(args.pmin), np.log(args.pmax), (n, k)))
        s = np.where(rng.random((n, k)) < args.peak_frac, peaks[:k] * np.exp(rng.normal(0, 5e-4, (n, k))), s)
        per_k.append((s, rng.gamma(0.5, 1.0, n)))
    post.append(per_k)
pky = rng.dirichlet(np.ones(args.nmod))
nu, nua, nub = fip.frequency_grid(args.pmin, args.pmax, args.tobs)

fip.fip_periodogram(post[:1], pky, nua, nub)                       # first call: context + code object load
t0 = time.perf_counter()
got, t = fip.fip_periodogram(post, pky, nua, nub, repeats=args.repeats, return_timing=True)
wall_rep = time.perf_counter() - t0
t0 = time.perf_counter()
fip.fip_periodogram(post, pky, nua, nub)
wall = time.perf_counter() - t0
periods, contrib, run_start = fip.flatten_posteriors(post, pky)
rows, np_max = periods.shape
valid = np.isfinite(periods)
f = 2 * np.pi / np.where(valid, periods, 1.0)
beg = np.searchsorted(nub, f, "right"); end = np.searchsorted(nua, f, "left")
updates = int(np.where(valid, np.maximum(end - beg, 0), 0).sum())
print(f"rows={rows} np_max={np_max} runs={args.runs} nfreq={nua.size} bin-updates(upper bound)={updates:.3e}")
print(f"GPU index kernel      {t['index_ms']:.4f} ms   ({rows * np_max / t['index_ms'] / 1e6:.1f} G searches/s x2)")
print(f"GPU accumulate kernel {t['accumulate_ms']:.4f} ms   ({rows / t['accumulate_ms'] / 1e3:.1f} M rows/s, "
      f"{updates / t['accumulate_ms'] / 1e6:.2f} G bin-updates/s)")
print(f"GPU host call (flatten + H2D + kernels + D2H) {wall * 1e3:.1f} ms")

if not args.no_cpu:
    from oracle import oracle, fip_oracle
    t0 = time.perf_counter()
    want = oracle.fip_accumulate(nua, nub, periods, contrib, run_start)
    c_s = time.perf_counter() - t0
    print(f"C fold (1 core)       {c_s * 1e3:.1f} ms   identical={np.array_equal(got, want)}")
    sub = [[None] + [(s[:2000], w[:2000]) for (s, w) in per_k[1:]] for per_k in post[:1]]
    t0 = time.perf_counter()
    fip_oracle.accumulate(sub, pky, nua, nub)
    py_s = time.perf_counter() - t0
    nsub = sum(len(p[0]) for p in sub[0][1:])
    print(f"reference-style numpy loop {py_s / nsub * 1e6:.1f} us/sample -> {py_s / nsub * rows:.1f} s for this input "
          f"(timed on {nsub} samples)")
