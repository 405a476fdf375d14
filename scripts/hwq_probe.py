#!/usr/bin/env python3
"""Does librvll's own request for eight hardware queues (its load-time constructor, csrc/rvll_comm.hip) take effect?  The streamed
262144-row cube -> theta -> log-L call, three ways, each in a fresh process:  python scripts/hwq_probe.py [library|python|nobody]
  library  librvll.so is loaded with GPU_MAX_HW_QUEUES unset, before evidence_amd is imported: only the constructor can have set it
  python   the Python binding sets it before it loads the library (rounds 3's way)
  nobody   RVLL_KEEP_HW_QUEUES=1: the runtime's default of four"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1] if len(sys.argv) > 1 else "library"
os.environ.pop("GPU_MAX_HW_QUEUES", None)
if mode == "nobody":
    os.environ["RVLL_KEEP_HW_QUEUES"] = "1"
if mode == "library":
    ctypes.CDLL(os.path.join(ROOT, "evidence_amd", "librvll.so"), mode=ctypes.RTLD_GLOBAL)
    os.environ["RVLL_KEEP_HW_QUEUES"] = "1"          # the binding then leaves the variable alone (the constructor has run already)
import numpy as np
from evidence_amd import GpuRVModel
from evidence_amd.synthetic import make_workload
w = make_workload(3)
with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
    n = 262144
    for _ in range(12):
        m.prior_loglike_batch(w.sample_cube(n, seed=3))
    ts = []
    for _ in range(10):
        cube = w.sample_cube(n, seed=4)
        t0 = time.perf_counter(); m.prior_loglike_batch(cube); ts.append(time.perf_counter() - t0)
    info = GpuRVModel.runtime_info()
    print(f"{mode:8s}: {np.median(ts) * 1e3:.2f} ms median, {min(ts) * 1e3:.2f} best per 262144-row call; GPU_MAX_HW_QUEUES {info['gpu_max_hw_queues_env']!r} set by {info['gpu_max_hw_queues_set_by']}")
