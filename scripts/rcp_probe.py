import sys; sys.path.insert(0, '.')
import numpy as np
from evidence_amd import GpuRVModel
from evidence_amd.synthetic import make_workload
w = make_workload(1)
with GpuRVModel(w.fixedpardict, w.table, w.parnames) as m:
    rng = np.random.default_rng(0); n = 8_000_000
    den = rng.uniform(0.01, 1.99, n); num = rng.normal(0, 1, n) * 10.0 ** rng.integers(-8, 3, n)
    rcp = m.debug_eval(11, den); err = np.abs(rcp * den - 1.0)
    print("v_rcp_f64 max rel err %.3e (2^%.1f)" % (err.max(), np.log2(err.max())))
    one = m.debug_eval(10, num, den); ieee = num / den
    bad = one != ieee
    print("div_1nr mismatches vs IEEE: %d of %d; max ulp diff %.2f" % (bad.sum(), n, (np.abs(one - ieee) / np.spacing(np.abs(ieee))).max()))
    fast = m.debug_eval(4, num, den); badf = fast != ieee
    print("div_fast mismatches: %d (%.2f%%); max ulp %.2f" % (badf.sum(), 100 * badf.mean(), (np.abs(fast - ieee) / np.spacing(np.abs(ieee))).max()))
