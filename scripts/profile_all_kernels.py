#!/usr/bin/env python3
"""Exercise every kernel of the library once with enough launches for a rocprofv3 summary (rocprof
evidence for the kernels besides the headline one): prior_kernel, prior_heavy_kernel, the one-launch cube -> log-L
kernel, the scalar-call server, the proposal walk, and the log-L kernel at the cfg4 / cfg5 shard sizes.

    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 scripts/profile_all_kernels.py
"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from evidence_amd import GpuRVModel  # noqa: E402
from evidence_amd.callbacks import make_ultranest_callbacks, wrapped_params  # noqa: E402
from evidence_amd.nested import run_nested_slice  # noqa: E402
from evidence_amd.synthetic import make_workload  # noqa: E402


def main():
    light = "--light" in sys.argv                 # PMC passes serialise kernels: fewer launches
    rep = 1 if light else 4
    w = make_workload(3)
    B = 16384
    with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
        m.dev_fill_cube(B, seed=5)
        for _ in range(50 * rep):                 # prior_kernel + prior_heavy_kernel + log-L (CU-wide form)
            m.dev_prior(B)
            m.dev_loglike(B)
        m.dev_sync()
        small = 2048
        m.dev_fill_cube(small, seed=6)
        for _ in range(100 * rep):                # loglike_kernel<0, 1>: the one-launch form (slim prior stage)
            m.dev_prior_loglike(small)
            m.dev_sync()
        m.set_kernel_form("tile")
        theta = w.sample_theta(B, seed=1)
        m.dev_upload_theta(theta)
        for _ in range(50 * rep):                 # loglike_kernel<0, 0>: the 256-thread tiles at the headline size
            m.dev_loglike(B)
        m.dev_sync()
        m.set_kernel_form("auto")
        x0 = theta[0]
        m.scalar_server(True)
        for _ in range(500 * rep):                # scalar_server_kernel (one persistent launch answers them all)
            m.log_likelihood(x0)
        m.scalar_server(False)
        vprior, vloglike = make_ultranest_callbacks(m, vectorized=True)
        t0 = time.perf_counter()
        ns = run_nested_slice(vprior, vloglike, m.ndim, nlive=16384, kbatch=8192, dlogz=1e-9,
                              max_calls=(6_000_000 if light else 30_000_000), wrapped=wrapped_params(m.parnames), seed=1,
                              prior_loglike=m.prior_loglike_batch, walker=m.slice_walk)   # slice_walk_kernel<0, false>
        print(f"nested sampling: {ns.ncall} calls in {time.perf_counter() - t0:.2f} s", file=sys.stderr)
        # round 3: the same run with the live set resident on the device (rvll_live_step: gather / scatter / moments kernels)
        t0 = time.perf_counter()
        ns = run_nested_slice(None, None, m.ndim, nlive=32768, kbatch=16384, dlogz=1e-9,
                              max_calls=(6_000_000 if light else 30_000_000), wrapped=wrapped_params(m.parnames), seed=1, live=m)
        print(f"nested sampling, resident live set: {ns.ncall} calls in {time.perf_counter() - t0:.2f} s", file=sys.stderr)
    # prior_heavy_kernel on a shape whose table fails its check: Beta(0.1, 0.2) -> full solver per element
    from evidence_amd.priors import PriorSpec, prior_constructor     # noqa: E402
    wr = make_workload(3)
    for n in (1, 2, 3):
        wr.input_dict[f"planet{n}"]["ecc"][2] = ["Beta", 0.1, 0.2]
    with GpuRVModel(wr.fixedpardict, wr.table, wr.parnames, priordict=wr.priordict()) as m:
        info = m.prior_table_info()
        print("Beta(0.1, 0.2) tables (measured error, evaluated by interpolation):", info, file=sys.stderr)
        m.dev_fill_cube(B, seed=8)
        for _ in range(20 * rep):
            m.dev_prior(B)
        m.dev_sync()
        t0 = time.perf_counter()
        for _ in range(20):
            m.dev_prior(B)
        m.dev_sync()
        print(f"prior transform with rejected tables: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us per {B} points "
              f"({B * 20 / (time.perf_counter() - t0):.3e} points/s)", file=sys.stderr)
    for cfg, b, n in ((4, 8192, 100 * rep), (5, 16384, 25 * rep)):      # the 8-GPU shard sizes of cfg4 / cfg5
        wk = make_workload(cfg)
        with GpuRVModel(wk.fixedpardict, wk.table, wk.parnames) as m:
            m.dev_upload_theta(wk.sample_theta(b, seed=2))
            for _ in range(n):
                m.dev_loglike(b)
            m.dev_sync()


if __name__ == "__main__":
    main()
