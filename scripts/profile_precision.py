#!/usr/bin/env python3
"""The log-L kernel at the cfg5 shard (5 planets + drift, 2000 epochs, 16384 points) in its three arithmetic modes, enough
launches of each for a rocprofv3 summary (rocprof + SQ counters of the reduced-precision instantiations):
loglike_cu_kernel<0, ...> (fp64, the parity mode), <1, ...> (mixed), <2, ...> (fp32).

    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 scripts/profile_precision.py [--light]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from evidence_amd import GpuRVModel  # noqa: E402
from evidence_amd.synthetic import make_workload  # noqa: E402


def main():
    light = "--light" in sys.argv
    n = 8 if light else 40
    w = make_workload(5)
    b = 16384
    theta = w.sample_theta(b, seed=2)
    for precision in ("fp64", "mixed", "fp32"):
        with GpuRVModel(w.fixedpardict, w.table, w.parnames, precision=precision) as m:
            m.dev_upload_theta(theta)
            t0 = time.perf_counter()
            while True:                             # launches for the summary; without counters also until the clocks are up
                for _ in range(n):
                    m.dev_loglike(b)
                m.dev_sync()
                if light or time.perf_counter() - t0 > 0.5:
                    break
            ms = m.dev_time_loglike(b, warmup=3, iters=20)
            print(f"cfg5 shard, {precision}: {ms} ms per launch (HIP events)", file=sys.stderr)


if __name__ == "__main__":
    main()
