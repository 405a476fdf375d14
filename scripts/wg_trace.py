#!/usr/bin/env python3
"""Where does one log-L launch spend its time?  ("21 % of each launch is ramp/tail".)

Runs the stamped twin of the fp64 kernel (rvll_dev_trace_loglike) on a BASELINE config and prints
  * the launch's span (first start -> last end) against the mean busy time of a workgroup slot,
  * how many workgroups are resident over time (ramp, plateau, tail),
  * workgroup durations: histogram, prologue share, wave imbalance inside a workgroup,
  * duration against the largest clamped eccentricity among the workgroup's points (Newton step counts grow
    with e: evidence/rvmodel/trueanomaly.c:17-33),
  * per-CU finishing times.

    python scripts/wg_trace.py [--config 3] [--batch 0] [--pb 0] [--out gpurun_out/wg_trace.npz]
"""
import argparse
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))

from evidence_amd import GpuRVModel  # noqa: E402
from evidence_amd.synthetic import CONFIGS, make_workload  # noqa: E402

TICK_US = 0.01          # s_memrealtime: 100 MHz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--pb", type=int, default=0)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--form", default="auto", choices=["auto", "tile", "cu"])
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    w = make_workload(args.config)
    B = args.batch or (CONFIGS[args.config]["batch"] // (8 if args.config in (4, 5) else 1))
    theta = w.sample_theta(B, seed=1234)
    model = GpuRVModel(w.fixedpardict, w.table, w.parnames)
    if args.pb:
        model.set_points_per_block(args.pb)
    model.set_kernel_form(args.form)
    model.dev_upload_theta(theta)
    tm = model.dev_time_loglike(B, warmup=50, iters=200)
    print(f"cfg{args.config}  B={B}  Ne={w.table.n_epochs}  event-timed kernel: mean {tm['kernel_ms_mean']*1e3:.2f} us, "
          f"min {tm['kernel_ms_min']*1e3:.2f} us; PB={tm['points_per_block']} blocks={tm['blocks']} threads={tm['threads']}")

    ecc_cols = [i for i, n in enumerate(w.parnames) if n.endswith("_ecc")]
    emax_pt = np.minimum(theta[:, ecc_cols].max(axis=1), 0.99) if ecc_cols else np.zeros(B)

    for rep in range(args.repeat):
        tr, pb = model.dev_trace_loglike(B, warmup=100)
        nb = tr.shape[0]
        t0 = tr[:, 0].astype(np.int64)
        base = t0.min()
        start = (t0 - base) * TICK_US
        dec = (tr[:, 1].astype(np.int64) - base) * TICK_US
        wend = (tr[:, 2:6].astype(np.int64) - base) * TICK_US
        end = (tr[:, 6].astype(np.int64) - base) * TICK_US
        hw = (tr[:, 7] & np.uint64(0xffffffff)).astype(np.int64)
        xcc = (tr[:, 7] >> np.uint64(32)).astype(np.int64) & 0xf
        cu_key = xcc * 65536 + (hw & 0xff00)            # se[15:13] sh[12] cu[11:8]
        dur = end - start
        span = end.max()
        if tm["threads"] != 256:          # CU-wide form: different stamps (rvll_kernels.h)
            rel = lambda k: (tr[:, k].astype(np.int64) - base) * TICK_US
            ld, st, de, w0, al = rel(1), rel(2), rel(3), rel(4), rel(5)
            print(f"\n--- CU-wide trace {rep}: {nb} workgroups x {pb} points, span {span:.2f} us; per workgroup (mean / max): "
                  f"start {start.mean():.2f}/{start.max():.2f}  theta landed +{np.mean(ld - start):.2f}/{np.max(ld - start):.2f}  "
                  f"staged +{np.mean(st - ld):.2f}/{np.max(st - ld):.2f}  decoded +{np.mean(de - st):.2f}/{np.max(de - st):.2f}  "
                  f"wave 0 out of items +{np.mean(w0 - de):.2f}/{np.max(w0 - de):.2f}  all out +{np.mean(al - w0):.2f}/{np.max(al - w0):.2f}  "
                  f"reduce+write +{np.mean(end - al):.2f}/{np.max(end - al):.2f}  duration {dur.mean():.2f}/{dur.max():.2f}")
            cu_end = np.array([end[cu_key == k].max() for k in np.unique(cu_key)])
            print(f"    per-CU last end: min {cu_end.min():.2f}  median {np.median(cu_end):.2f}  max {cu_end.max():.2f} us; "
                  f"first-round workgroups (start < 2 us): {(start < 2).sum()}; later ones start {np.mean((start[start >= 2]) if (start >= 2).any() else [0]):.2f} on average")
            continue
        print(f"\n--- trace {rep}: {nb} workgroups x {pb} points, span {span:.2f} us "
              f"(first start 0, last start {start.max():.2f}, first end {end.min():.2f}, last end {end.max():.2f})")
        ncu = len(np.unique(cu_key))
        busy = dur.sum()
        print(f"CUs seen {ncu}; sum of workgroup durations {busy:.0f} us = {busy / ncu:.2f} us per CU "
              f"-> mean residency {busy / ncu / span:.2f} workgroups per CU over the span")
        print(f"workgroup duration: mean {dur.mean():.2f}  p5 {np.percentile(dur, 5):.2f}  median {np.median(dur):.2f}  "
              f"p95 {np.percentile(dur, 95):.2f}  max {dur.max():.2f} us")
        print(f"prologue (stage + decode): mean {np.mean(dec - start):.2f} us = {100 * np.mean(dec - start) / dur.mean():.1f} % of a workgroup")
        wspread = wend.max(axis=1) - wend.min(axis=1)
        print(f"item loop, last window: wave end spread inside a workgroup mean {wspread.mean():.2f}  p95 {np.percentile(wspread, 95):.2f}  max {wspread.max():.2f} us")
        print(f"epilogue (after slowest wave): mean {np.mean(end - wend.max(axis=1)):.2f} us")
        # residency timeline
        edges = np.arange(0.0, span + 2.0, 2.0)
        print("resident workgroups (mean over 2 us bins):")
        line = []
        for lo in edges[:-1]:
            hi = lo + 2.0
            ov = np.clip(np.minimum(end, hi) - np.maximum(start, lo), 0, None).sum() / 2.0
            line.append(f"{ov:.0f}")
        print("  " + " ".join(line))
        # second-wave starts: when do slots free up?
        first = np.sort(start)[: min(nb, 1024)]
        print(f"dispatch ramp: 1st..1024th start {first[0]:.2f} .. {first[-1]:.2f} us")
        # per CU finishing
        cu_end = np.array([end[cu_key == k].max() for k in np.unique(cu_key)])
        cu_n = np.array([(cu_key == k).sum() for k in np.unique(cu_key)])
        print(f"per-CU last end: min {cu_end.min():.2f}  median {np.median(cu_end):.2f}  max {cu_end.max():.2f} us; "
              f"workgroups per CU min {cu_n.min()} max {cu_n.max()}")
        idle_tail = (span - cu_end).mean()
        print(f"mean idle tail per CU {idle_tail:.2f} us = {100 * idle_tail / span:.1f} % of the span")
        # per XCD
        for x in np.unique(xcc):
            m = xcc == x
            print(f"  XCC {x}: {m.sum()} workgroups, last end {end[m].max():.2f}, mean duration {dur[m].mean():.2f}")
        # duration vs eccentricity
        share = pb if tm["threads"] == 256 else -(-B // nb)       # points per workgroup
        emax_wg = np.array([emax_pt[i * share:(i + 1) * share].max() if i * share < B else 0.0 for i in range(nb)])
        print("duration against the largest clamped eccentricity of the workgroup's points:")
        for lo, hi in ((0, .5), (.5, .7), (.7, .8), (.8, .9), (.9, .95), (.95, .98), (.98, 1.0)):
            m = (emax_wg >= lo) & (emax_wg < hi)
            if m.any():
                print(f"  e_max in [{lo:.2f},{hi:.2f}): {m.sum():5d} workgroups, duration mean {dur[m].mean():6.2f}  max {dur[m].max():6.2f} us, "
                      f"mean start {start[m].mean():6.2f}")
        order = np.argsort(-dur)[:10]
        print("ten longest workgroups: " + ", ".join(f"#{i} {dur[i]:.1f}us e={emax_wg[i]:.3f} start={start[i]:.1f}" for i in order))
        late = np.argsort(-end)[:10]
        print("ten last to end:       " + ", ".join(f"#{i} end={end[i]:.1f} dur={dur[i]:.1f} e={emax_wg[i]:.3f}" for i in late))
        if args.out and rep == 0:
            Path(args.out).parent.mkdir(parents=True, exist_ok=True)
            np.savez_compressed(args.out, trace=tr, pb=pb, emax_wg=emax_wg)
    model.close()


if __name__ == "__main__":
    main()
