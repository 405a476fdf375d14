#!/usr/bin/env python3
"""Soak of the host-buffer entry points across their size classes (run on the GPU box).  rvll_loglike_batch,
rvll_prior_batch and rvll_prior_loglike_batch pick a transport by the size of the call — scalar server, zero-copy
through the mapped pinned blocks, pinned landing zone for the results, overlapped chunks on two streams, staged
downloads (evidence_amd/csrc/rvll_api.hip) — and the rows of a batch are independent, so every call on a slice of
one big table must return exactly the bits of the same rows computed once through the device-resident path.
Sizes: every threshold the library has, one row either side of it, then random sizes for --seconds."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evidence_amd import GpuRVModel
from evidence_amd.synthetic import make_workload

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=60.0)
ap.add_argument("--cfgs", default="3,5,2")
args = ap.parse_args()
rng = np.random.default_rng(2024)
NBIG = 300_000


def edges(D):
    out = {1, 2, 3, 63, 64, 65, 511, 512, 513, 4095, 4096, 4097, 16383, 16384, 16385, 32767, 32768, 32769,
           65535, 65536, 65537, 131071, 131072, 131073, 262144, NBIG}
    for T in (64 << 10, 384 << 10, 1 << 20, 8 << 20, 24 << 20, 32 << 20):
        for r in (8 * D, 16 * D, 12, 16, 8 * D + 12, 16 * D + 12, 8):
            n = T // r
            out.update(k for k in (n - 1, n, n + 1) if 1 <= k <= NBIG)
    return sorted(out)


def check(m, n, o, ref, what):
    cube, theta, logl, flags = (a[o:o + n] for a in ref)
    if what == 0:
        got, fl = m.log_likelihood_batch(theta, return_flags=True)
        assert np.array_equal(got, logl) and np.array_equal(fl, flags), ("loglike_batch", n, o)
    elif what == 1:
        assert np.array_equal(m.prior_transform_batch(cube), theta), ("prior_batch", n, o)
    else:
        th, got, fl = m.prior_loglike_batch(cube, return_flags=True)
        assert np.array_equal(th, theta) and np.array_equal(got, logl) and np.array_equal(fl, flags), ("prior_loglike_batch", n, o)


t_end_all = time.time() + args.seconds
cfgs = [int(c) for c in args.cfgs.split(",")]
total = 0
for ci, cfg in enumerate(cfgs):
    w = make_workload(cfg)
    cube = w.sample_cube(NBIG, seed=40 + cfg)
    cube[:: 997] = np.clip(cube[:: 997] * 1e-9, 0.0, 1.0)           # a few rows hard against the cube's walls
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        # the reference table: device-resident path, one piece
        m.dev_upload_cube(cube); m.dev_prior(NBIG); m.dev_loglike(NBIG)
        theta, logl, flags = m.dev_download(NBIG, theta=True, flags=True)
        theta, logl, flags = theta.copy(), logl.copy(), flags.copy()
        ref = (cube, theta, logl, flags)
        ncalls = 0
        for n in edges(m.ndim):
            o = int(rng.integers(0, NBIG - n + 1))
            for what in (0, 1, 2):
                check(m, n, o, ref, what); ncalls += 1
        print(f"cfg{cfg} (D={m.ndim}): {len(edges(m.ndim))} threshold sizes x 3 entry points ok", flush=True)
        t_end = time.time() + max(5.0, (t_end_all - time.time()) / (len(cfgs) - ci))
        last = time.time()
        while time.time() < t_end:
            r = rng.random()
            n = int(rng.integers(1, 8) if r < 0.15 else rng.integers(1, 8000) if r < 0.6 else rng.integers(8000, 100000) if r < 0.93
                    else rng.integers(100000, NBIG))
            o = int(rng.integers(0, NBIG - n + 1))
            if rng.random() < 0.1:
                m.scalar_server(bool(rng.random() < 0.5))
            check(m, n, o, ref, int(rng.integers(0, 3))); ncalls += 1
            if rng.random() < 0.05:                                 # the scalar forms in between
                k = int(rng.integers(0, NBIG))
                assert m.log_likelihood(theta[k]) == logl[k]
                assert np.array_equal(m.prior_transform(cube[k]), theta[k])
            if time.time() - last > 20:
                last = time.time()
                print(f"... cfg{cfg}: {ncalls} calls", flush=True)
        total += ncalls
        print(f"cfg{cfg}: {ncalls} calls, all bit-identical to the resident path", flush=True)
print(f"host-call soak ok: {total} calls over cfgs {cfgs}")
