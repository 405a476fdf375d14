#!/usr/bin/env python3
"""How large are the Newton steps whose end point the next sin/cos is taken at?  (CPU only, numpy.)

The solver (evidence/rvmodel/trueanomaly.c:17-33) starts at E = M and stops at |dE| <= tol.  Every iteration but the
first evaluates sin/cos at E_prev + dE; if |dE| is small for EVERY lane of a wave, the pair can be rotated from the
previous one by a short polynomial in dE instead of a fresh range reduction + two degree-13 polynomials.  This prints,
per BASELINE config, the share of wave-level sin/cos evaluations that fall under a few bounds H (a wave = 64
consecutive (point, epoch) items, as the tile kernel forms them).

    python scripts/newton_step_stats.py [--config 3] [--points 1024]
"""
import argparse
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from evidence_amd.synthetic import make_workload  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--points", type=int, default=1024)
    args = ap.parse_args()
    w = make_workload(args.config)
    th = w.sample_theta(args.points, seed=1234)
    t = w.table.time
    names = list(w.parnames)
    planets = sorted({n.split("_")[0] for n in names if n.startswith("planet")})
    bounds = [2.0 ** -k for k in (6, 5, 4, 3, 2)]
    tol, itmax = 1e-4, 1_000_000
    tot_eval = 0
    under = np.zeros(len(bounds))
    mixed = np.zeros(len(bounds))
    lane_iters = 0
    wave_iters = 0
    for p in planets:
        col = {k: names.index(f"{p}_{k}") for k in ("period", "ecc") if f"{p}_{k}" in names}
        P = th[:, col["period"]][:, None]
        e = np.minimum(th[:, col["ecc"]], 0.99)[:, None]
        rng = np.random.default_rng(1)
        M = (2 * np.pi / P * (t[None, :] - t[0]) + rng.uniform(0, 2 * np.pi, (len(th), 1))).reshape(-1)
        ee = np.broadcast_to(e, (len(th), len(t))).reshape(-1)
        n = (M.size // 64) * 64
        M, ee = M[:n].reshape(-1, 64), ee[:n].reshape(-1, 64)
        E = M.copy()
        active = np.ones(M.shape, bool)
        prev = np.full(M.shape, np.inf)                  # |dE| that led to the current E (first evaluation: none)
        for it in range(60):
            wave_on = active.any(axis=1)
            if not wave_on.any():
                break
            # a wave-level evaluation happens for every wave with an active lane
            mx = np.where(active, prev, 0.0).max(axis=1)[wave_on]
            tot_eval += mx.size
            mn = np.where(active, prev, np.inf).min(axis=1)[wave_on]
            for k, H in enumerate(bounds):
                under[k] += (mx <= H).sum()
                mixed[k] += ((mn <= H) & (mx > H)).sum()
            wave_iters += wave_on.sum()
            lane_iters += active.sum()
            f = E - ee * np.sin(E) - M
            fp = 1 - ee * np.cos(E)
            dE = -f / fp
            E = np.where(active, E + dE, E)
            prev = np.where(active, np.abs(dE), prev)
            active &= np.abs(dE) > tol
    print(f"cfg{args.config}: {len(planets)} planets, {len(t)} epochs, {args.points} points; mean Newton steps per lane "
          f"{lane_iters / (M.size * len(planets)):.2f}, per wave {wave_iters / (M.shape[0] * len(planets)):.2f}")
    for H, u, m in zip(bounds, under, mixed):
        print(f"  wave-level sin/cos evaluations with every active lane's |dE| <= {H:.4f}: {100 * u / tot_eval:5.1f} %"
              f"   (some lanes under, some over: {100 * m / tot_eval:5.1f} %)")


if __name__ == "__main__":
    main()
