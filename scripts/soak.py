#!/usr/bin/env python3
"""Sustained-load soak (run on the GPU box): ~30 s of device-resident launches cycling the pipeline lanes, interleaved
with prior transforms, uploads and host-buffer calls; every result must be bit-identical to the first one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evidence_amd import GpuRVModel
from evidence_amd.synthetic import make_workload

w = make_workload(3)
B = 16384
theta = w.sample_theta(B, seed=5)
cube = w.sample_cube(B, seed=6)
with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
    want = m.log_likelihood_batch(theta)
    want_theta, want_pl = m.prior_loglike_batch(cube)
    m.comm_init(GpuRVModel.comm_unique_id(), 1, 0)
    t0, rounds, launches = time.time(), 0, 0
    while time.time() - t0 < float(sys.argv[1]) if len(sys.argv) > 1 else time.time() - t0 < 30:
        m.dev_upload_theta(theta)
        for _ in range(200):
            m.dev_loglike(B); m.allgather_logl(B); launches += 1
        assert np.array_equal(m.download_gathered(B), want), "gathered log-L changed"
        assert np.array_equal(m.dev_download(B)[1], want), "resident log-L changed"
        m.dev_upload_cube(cube); m.dev_prior(B); m.dev_loglike(B)
        th, ll, _ = m.dev_download(B, theta=True)
        assert np.array_equal(th, want_theta) and np.array_equal(ll, want_pl), "prior+loglike changed"
        assert np.array_equal(m.log_likelihood_batch(theta[:777]), want[:777]), "host call changed"
        assert m.log_likelihood(theta[5]) == want[5]
        rounds += 1
    m.comm_destroy()
print(f"soak ok: {rounds} rounds, {launches} pipelined launches + gathers in {time.time() - t0:.1f} s, all results bit-identical")
