#!/usr/bin/env python3
"""Large theta downloads (rvll_dev_download): into a fresh numpy array per call against into one that is reused.  A
device-to-host copy into a range the runtime has pinned before runs at 50 GB/s; into a fresh mapping — what every numpy
array above glibc's 32 MB mmap threshold is — at 1.5 GB/s, which is why the library stages such rows through its pinned
blocks from 32 MB on.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from evidence_amd import GpuRVModel, _abi
from evidence_amd.synthetic import make_workload

w = make_workload(3)
with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
    for n in (65536, 131072, 200000, 262144, 400000):
        cube = w.sample_cube(n, 1)
        m.dev_reserve(n)
        m.dev_upload_cube(cube)
        m.dev_prior(n)
        m.dev_sync()
        mb = n * m.ndim * 8 / 2 ** 20

        def timed(f, reps=6):
            f()
            t0 = time.perf_counter()
            for _ in range(reps):
                f()
            return (time.perf_counter() - t0) / reps * 1e6

        fresh = timed(lambda: m.dev_download(n, theta=True, logl=False))
        keep = np.empty((n, m.ndim))
        keep[:] = 0.0
        reuse = timed(lambda: _abi.check(m._lib.rvll_dev_download(m._h, n, _abi.as_dp(keep), None, None)))
        t0 = time.perf_counter()
        for _ in range(6):
            a = np.empty((n, m.ndim)); a[:] = 0.0
        touch = (time.perf_counter() - t0) / 6 * 1e6
        print(f"n={n:7d} ({mb:5.1f} MB of theta): download into a fresh array {fresh:9.1f} us, into a reused one {reuse:8.1f} us "
              f"({mb / 1024 / (reuse * 1e-6):5.1f} GB/s); allocating and touching such an array alone {touch:9.1f} us", flush=True)
