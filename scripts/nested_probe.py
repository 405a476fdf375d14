#!/usr/bin/env python3
"""bench.py's nested-sampling leg by itself (cfg3, 32768 live points, 16384 deaths per iteration, live set resident on
the device), once per setting of the walk's switches: likelihood calls per second inside rvll_live_step and end to end.

    python scripts/nested_probe.py [name=ENV1:val,ENV2:val ...]      (run on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evidence_amd import GpuRVModel
from evidence_amd.callbacks import wrapped_params
from evidence_amd.nested import run_nested_slice
from evidence_amd.synthetic import make_workload

SWITCHES = ("RVLL_WALK_ROUNDS", "RVLL_ROUNDS_PER", "RVLL_WALK_CR", "RVLL_ROUNDS_GROUPS", "RVLL_ROUNDS_FREE", "RVLL_ROUNDS_DEPTH", "RVLL_WALK_SPEC", "RVLL_ROUNDS_W", "RVLL_ROUNDS_PB", "RVLL_ROUNDS_MODE", "RVLL_ROUNDS_FORM", "RVLL_ROUNDS_PRIO", "RVLL_ROUNDS_CHAIN")
settings = [("single-kernel", {"RVLL_WALK_ROUNDS": "0"}), ("rounds default", {})]
for arg in sys.argv[1:]:
    name, _, rest = arg.partition("=")
    settings.append((name, dict(kv.split(":") for kv in rest.split(",") if kv)))
verbose = bool(os.environ.get("PROBE_VERBOSE"))
nlive, kbatch = int(os.environ.get("PROBE_NLIVE", "32768")), int(os.environ.get("PROBE_KBATCH", "16384"))
w = make_workload(3)
with GpuRVModel(w.fixedpardict, w.table, w.parnames, priordict=w.priordict()) as m:
    kw = dict(nlive=nlive, kbatch=kbatch, dlogz=1e-9, max_calls=int(os.environ.get("PROBE_CALLS", "60000000")),
              wrapped=wrapped_params(m.parnames), seed=1)
    for name, env in settings:
        for k in SWITCHES:
            os.environ.pop(k, None)
        os.environ.update(env)
        best = None
        for rep in range(2):
            inside = {"s": 0.0, "calls": 0, "slots": 0, "rounds": 0, "it": 0}

            class Live:
                live_init, live_get, live_dead, live_dead_count = m.live_init, m.live_get, m.live_dead, m.live_dead_count
                if not os.environ.get('PROBE_HOST_SORT'):
                    live_sort = m.live_sort

                @staticmethod
                def live_step(*a, **k):
                    t2 = time.perf_counter()
                    res = m.live_step(*a, **k)
                    dt = time.perf_counter() - t2
                    inside["s"] += dt
                    inside["calls"] += res[1]
                    inside["slots"] += m.slice_walk_evaluated()
                    inside["rounds"] += m.slice_walk_rounds()
                    inside["it"] += 1
                    if verbose and rep == 0:
                        print(f"    iteration {inside['it']:3d}: {res[1]:8d} calls, {m.slice_walk_evaluated():8d} slots, {m.slice_walk_rounds():5d} rounds, "
                              f"{dt * 1e3:7.2f} ms = {res[1] / dt:.3e}/s", flush=True)
                    return res

            t1 = time.perf_counter()
            ns = run_nested_slice(None, None, m.ndim, live=Live, **kw)
            el = time.perf_counter() - t1
            cur = (inside["calls"] / inside["s"], ns.ncall / el, ns.ncall, inside["slots"] / max(1, inside["calls"]), inside["rounds"], inside["it"], ns.logz, ns.timing, el, inside["s"])
            if best is None or cur[0] > best[0]:
                best = cur
        print(f"{name:28s}: inside the step {best[0]:.3e} calls/s, end to end {best[1]:.3e}  ({best[2]} calls, {best[3]:.3f} slots per call, "
              f"{best[4]} rounds over {best[5]} iterations, ln Z so far {best[6]:.3f}; per iteration: order {best[7]['order_s'] / best[7]['turns'] * 1e3:.3f} ms, "
              f"live step {best[9] / best[5] * 1e3:.3f} ms, everything else incl. set-up and final downloads {(best[8] - best[9] - best[7]['order_s']) / best[5] * 1e3:.3f} ms)", flush=True)
