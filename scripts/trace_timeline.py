#!/usr/bin/env python3
"""Timeline figures from a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv): per kernel name count / mean / total duration,
the share of the traced span in which at least one kernel runs, and the gaps between consecutive kernels of one queue.

    python scripts/trace_timeline.py <kernel_trace.csv> [name-substring to restrict the span to]"""
import csv, sys, re
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
pick = sys.argv[2] if len(sys.argv) > 2 else None
ev = []
for r in rows:
    name = re.sub(r"rvll::\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
    name = re.sub(r"^void ", "", name)
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "0"), r.get("Stream_Id", "0")))
ev.sort()
if pick:
    sel = [e for e in ev if pick in e[2]]
    lo, hi = sel[0][0], sel[-1][1]
    ev = [e for e in ev if e[0] >= lo and e[1] <= hi]
span = ev[-1][1] - ev[0][0]
by = defaultdict(list)
for s, e, n, q, st in ev:
    by[n].append(e - s)
print(f"{len(ev)} dispatches over {span / 1e3:.1f} us")
for n, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {n[:70]:70s} n={len(d):6d} mean {sum(d) / len(d) / 1e3:8.2f} us  total {sum(d) / 1e3:10.1f} us ({100 * sum(d) / span:5.1f} % of the span)")
# union of busy intervals
busy, cur_s, cur_e = 0, None, None
for s, e, *_ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"some kernel running {100 * busy / span:.1f} % of the span; idle {(span - busy) / 1e3:.1f} us")
# per-queue gaps
perq = defaultdict(list)
for s, e, n, q, st in ev:
    perq[q].append((s, e, n))
for q, lst in perq.items():
    gaps = defaultdict(list)
    for (s0, e0, n0), (s1, e1, n1) in zip(lst, lst[1:]):
        gaps[(n0[:28], n1[:28])].append(s1 - e0)
    print(f"queue {q}: {len(lst)} dispatches")
    for k, g in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:6]:
        g2 = sorted(g)
        print(f"    {k[0]:28s} -> {k[1]:28s} n={len(g):5d} gap mean {sum(g) / len(g) / 1e3:7.2f} us  median {g2[len(g2) // 2] / 1e3:7.2f}  total {sum(g) / 1e3:9.1f} us")
