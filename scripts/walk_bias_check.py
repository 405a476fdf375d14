#!/usr/bin/env python3
"""Bias check of the nested sampler with the walk driven from the host, on the device, and with the live set resident there: a 4-D Gaussian
likelihood built from the RV model itself (four instruments, one unit-variance datum each, free offsets,
Uniform(-10, 10) priors: ln Z = -4 ln 20), many seeds each.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evidence_amd import GpuRVModel, priors as P
from evidence_amd.callbacks import make_ultranest_callbacks
from evidence_amd.data import EpochTable
from evidence_amd.nested import run_nested_slice
# 4-D Gaussian: four instruments with one unit-variance datum each, offsets free, Uniform(-10,10): ln Z = -4 ln 20
names = ["a", "b", "c", "d"]
table = EpochTable.from_arrays(names, [1.0, 2.0, 3.0, 4.0], [0.3, -0.2, 0.1, 0.0], [1.0, 1.0, 1.0, 1.0], [0, 1, 2, 3])
pri = {f"{n}_offset": P.Uniform(-10, 10) for n in names}
truth = -4 * np.log(20.0)
with GpuRVModel({}, table, list(pri), priordict=pri) as m:
    prior, loglike = make_ultranest_callbacks(m, vectorized=True)
    for nlive, kb in ((400, 100), (2000, 500), (2000, 1000)):
        kw = dict(nlive=nlive, kbatch=kb, dlogz=0.01, max_calls=100_000_000, nsteps=12)
        host = [run_nested_slice(prior, loglike, 4, seed=s, prior_loglike=m.prior_loglike_batch, **kw).logz for s in range(1, 17)]
        dev = [run_nested_slice(prior, loglike, 4, seed=s, walker=m.slice_walk, **kw).logz for s in range(1, 33)]
        live = [run_nested_slice(None, None, 4, seed=s, live=m, **kw).logz for s in range(1, 33)]     # live set + whitening on the device
        for name, v in (("host", host), ("dev ", dev), ("live", live)):
            print(f"nlive={nlive} kbatch={kb} {name}: mean-truth {np.mean(v) - truth:+.4f} +- {np.std(v) / np.sqrt(len(v)):.4f} (sd {np.std(v):.3f}, n={len(v)})", flush=True)
