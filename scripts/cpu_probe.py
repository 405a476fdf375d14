import os, sys, time
sys.path.insert(0, os.getcwd())
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
from evidence_amd.synthetic import make_workload
from evidence_amd.layout import compile_layout
from oracle.oracle import OracleModel
w = make_workload(3); L = compile_layout(w.parnames, w.fixedpardict, w.table.insts)
om = OracleModel(L, w.table); th = w.sample_theta(16384, 1)
for nt in (1, 8, 16, 32, 64, 128):
    n = 2048 if nt == 1 else 16384
    om.loglike(th[:n], nthreads=nt); t=time.perf_counter(); om.loglike(th[:n], nthreads=nt); dt=time.perf_counter()-t
    print("threads", nt, "evals/s %.0f" % (n/dt))
