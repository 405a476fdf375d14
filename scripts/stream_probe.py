#!/usr/bin/env python3
"""Large host-buffer calls: the streamed route (chunks through pinned blocks, the host's copies on worker threads;
rvll_api.hip, stream_host_batch) against the routes it replaces (RVLL_STREAM_MIN / RVLL_STREAM_LOGLIKE: measurement
switches), with the caller's input array the same one every call (the runtime pins and caches its pages) and a fresh
one every call (what a sampler does), and what a fresh 40 MB result array costs by itself.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evidence_amd import GpuRVModel
from evidence_amd.synthetic import make_workload

w = make_workload(3)
sizes = (32768, 65536, 131072, 262144, 524288)
print(f"host cpus {os.cpu_count()}  affinity {len(os.sched_getaffinity(0))}", flush=True)


def kept(f, x, reps):
    for _ in range(3):
        out = f(x)
    t0 = time.perf_counter()
    for _ in range(reps):
        out = f(x)
    return (time.perf_counter() - t0) / reps, out


def fresh(f, src, reps):
    tot = 0.0
    for r in range(reps + 2):
        x = src.copy()
        t0 = time.perf_counter(); f(x); dt = time.perf_counter() - t0
        tot += dt if r >= 2 else 0.0
        del x
    return tot / reps


t0 = time.perf_counter()
for _ in range(10):
    a = np.empty((262144, 19)); a[::512, 0] = 1.0; del a
print(f"np.empty((262144, 19)) + one write per page + free: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms "
      f"(results of 32 MB and more come from recycled blocks: evidence_amd/engine.py)", flush=True)

ref = {}
for n in sizes:
    cube, theta = w.sample_cube(n, 2), w.sample_theta(n, 1)
    for mode in ("old", "streamed", "streamed, 2 workers", "streamed, chunks of 32768"):
        if mode not in ("old", "streamed") and n != 262144:
            continue
        os.environ["RVLL_STREAM_MIN"] = str(1 << 40) if mode == "old" else "1"
        os.environ["RVLL_STREAM_LOGLIKE"] = "1"
        os.environ.pop("RVLL_COPY_THREADS", None); os.environ.pop("RVLL_STREAM_CHUNK", None)
        if "workers" in mode:
            os.environ["RVLL_COPY_THREADS"] = "2"
        if "chunks" in mode:
            os.environ["RVLL_STREAM_CHUNK"] = "32768"
        with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
            reps = max(5, min(40, (1 << 22) // n))
            dt, (th, ll) = kept(m.prior_loglike_batch, cube, reps)
            dtf = fresh(m.prior_loglike_batch, cube, reps)
            dt2, ll2 = kept(m.log_likelihood_batch, theta, reps)
            dt2f = fresh(m.log_likelihood_batch, theta, reps)
            ref.setdefault(n, (th.copy(), ll.copy(), ll2.copy()))
            same = bool(np.array_equal(th, ref[n][0]) and np.array_equal(ll, ref[n][1]) and np.array_equal(ll2, ref[n][2]))
            print(f"n={n:7d} {mode:26s} cube->theta->logL: same input {dt*1e3:6.3f} ms {n/dt:.3e}/s, fresh input {dtf*1e3:6.3f} ms {n/dtf:.3e}/s"
                  f"   theta->logL: same input {dt2*1e3:6.3f} ms {n/dt2:.3e}/s, fresh input {dt2f*1e3:6.3f} ms {n/dt2f:.3e}/s   same bits={same}", flush=True)
