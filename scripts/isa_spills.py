#!/usr/bin/env python3
"""v_readlane / v_writelane (scalar-register spill traffic) per loop nest depth of one kernel in a -save-temps .s file.
    python scripts/isa_spills.py file.s [kernel-substring]"""
import re, sys
path = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "loglike_cu_kernelILi0ELb0ELi0E"
s = open(path).read()
m = re.search(r'^(_ZN4rvll\S*' + re.escape(kern) + r'\S*):', s, re.M)
body = s[m.start():s.index('s_endpgm', m.start())].split('\n')
depth, stats = 0, {}
for l in body:
    if l.startswith('.LBB'):
        d = re.search(r'Depth=(\d+)', l)
        nxt = body[body.index(l) + 1] if False else ''
        depth = int(d.group(1)) if d else depth
        # a label line carries either "Loop Header: Depth=N" or "in Loop: Header=... Depth=N"; others follow in comment lines
        continue
    c = re.search(r';\s+(?:Parent Loop|in Loop|=>This).*Depth=(\d+)', l)
    if c and l.strip().startswith(';'):
        depth = max(depth, int(c.group(1))) if 'Parent' not in l else depth
        continue
    st = stats.setdefault(depth, [0, 0, 0, 0])
    if re.match(r'\s+v_readlane', l): st[0] += 1
    elif re.match(r'\s+v_writelane', l): st[1] += 1
    if re.match(r'\s+v_', l): st[2] += 1
    if re.match(r'\s+s_', l): st[3] += 1
for d in sorted(stats):
    print(f"depth {d}: readlane {stats[d][0]:4d}  writelane {stats[d][1]:4d}  VALU {stats[d][2]:5d}  SALU {stats[d][3]:5d}")
