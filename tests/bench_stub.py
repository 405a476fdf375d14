"""A stand-in for GpuRVModel's device-resident interface, for the CPU tests of bench.py's LAUNCH PATH only
(RVLL_BENCH_MODEL_FOR_TESTS=tests.bench_stub:StubModel): self-launch, rendezvous, socket all-gather, verification,
watchdog / lost-peer handling, the one JSON line.  It evaluates a closed-form stand-in for log-L with numpy — no
kernel, no oracle — and bench.py marks every line produced with it as not a measurement.

Failure injection (environment): RVLL_STUB_DIE_RANK=r and RVLL_STUB_DIE_AFTER=n make rank r leave with exit code 7
inside its n-th dev_loglike call."""
import os

import numpy as np

from evidence_amd.layout import compile_layout


class StubModel:
    _calls = 0

    @classmethod
    def device_count(cls):
        return int(os.environ.get("RVLL_STUB_DEVICES", "1"))

    @staticmethod
    def runtime_info():
        return {"stub": True}

    @staticmethod
    def comm_unique_id():
        raise RuntimeError("the stub has no RCCL")

    def __init__(self, fixedpardict, table, parnames, device=0, precision="fp64"):
        self.table = table
        self.layout = compile_layout(parnames, dict(fixedpardict), list(table.insts), [])
        self.parnames = list(self.layout.parnames)
        self.precision = precision
        self._theta = self._logl = None
        self._marks = [0.0, 0.0]

    @property
    def ndim(self):
        return len(self.parnames)

    def set_points_per_block(self, pb):
        pass

    def set_wander_exact(self, on=True):
        pass

    def dev_upload_theta(self, theta):
        self._theta = np.array(theta, dtype=np.float64)

    def dev_loglike(self, n):
        StubModel._calls += 1
        if (os.environ.get("RVLL_STUB_DIE_RANK") == os.environ.get("RANK", "0")
                and StubModel._calls >= int(os.environ.get("RVLL_STUB_DIE_AFTER", "0")) > 0):
            os._exit(7)
        th = self._theta[:n]
        v = -(th * th).sum(axis=1) - 1.0
        self._logl = v.astype(np.float32).astype(np.float64) if self.precision != "fp64" else v

    def dev_sync(self):
        pass

    def dev_download(self, n, theta=False, logl=True, flags=False):
        return (self._theta[:n].copy() if theta else None, self._logl[:n].copy() if logl else None,
                np.zeros(n, dtype=np.int32) if flags else None)

    def dev_mark(self, which):
        import time
        self._marks[which] = time.perf_counter()

    def dev_mark_elapsed_ms(self):
        return (self._marks[1] - self._marks[0]) * 1e3

    def dev_time_loglike(self, n, warmup=3, iters=20):
        return {"kernel_ms_mean": 1.0, "kernel_ms_min": 1.0, "kernel_ms_median": 1.0, "total_ms": float(iters),
                "evals": n * iters, "launches": iters, "points_per_block": 8, "blocks": (n + 7) // 8, "threads": 256}

    def comm_destroy(self):
        pass

    def close(self):
        pass
