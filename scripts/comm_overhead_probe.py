#!/usr/bin/env python3
"""Per-step cost of the all-gather machinery (event record, stream wait, RCCL all-gather on the comm stream,
log-L ping-pong) measured with a ONE-rank communicator on one GPU: steps with and without the gather."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evidence_amd import GpuRVModel
from evidence_amd.synthetic import make_workload

w = make_workload(3)
B = 16384
theta = w.sample_theta(B, seed=1)
with GpuRVModel(w.fixedpardict, w.table, w.parnames) as m:
    m.dev_upload_theta(theta)
    m.comm_init(GpuRVModel.comm_unique_id(), 1, 0)
    lanes = int(os.environ.get("RVLL_LANES", "3"))
    if lanes > 1:
        m.comm_set_lanes(m.comm_add_lanes(lanes))
    print(f"lanes requested {lanes}; runtime {GpuRVModel.runtime_info()}", flush=True)
    for gather in (False, True, False, True):
        for _ in range(300):
            m.dev_loglike(B)
            if gather:
                m.allgather_logl(B)
        m.dev_sync()
        t0 = time.perf_counter()
        K = 2000
        for _ in range(K):
            m.dev_loglike(B)
            if gather:
                m.allgather_logl(B)
        t_enq = (time.perf_counter() - t0) / K
        m.dev_sync()
        dt = (time.perf_counter() - t0) / K
        print(f"gather={gather!s:5}  {dt * 1e6:7.2f} us/step  {B / dt:.3e} evals/s   host enqueue {t_enq * 1e6:6.2f} us/step", flush=True)
    m.comm_destroy()
