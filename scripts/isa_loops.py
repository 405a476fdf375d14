#!/usr/bin/env python3
"""Instruction counts of the innermost loops of one kernel in the -save-temps ISA (make -C evidence_amd/csrc asm):
VALU / SALU / moves / lane reads per loop body — the Newton loop's 43 VALU instructions are checked with this."""
import re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "build/rvll_kernels-hip-amdgcn-amd-amdhsa-gfx950.s"
kern = sys.argv[2] if len(sys.argv) > 2 else "loglike_cu_kernelILi0ELb0ELi0E"
s = open(path).read()
i = s.index(next(l for l in s.split("\n") if kern in l and l.endswith(":") or (kern in l and ":" in l and l.startswith("_Z"))).split(":")[0] + ":")
body = s[i:s.index("s_endpgm", i)].split("\n")
labels = {l.split(":")[0]: k for k, l in enumerate(body) if l.startswith(".LBB")}
for lab, k in labels.items():
    hdr = "\n".join(body[k:k + 6])
    if "Inner Loop Header" not in hdr:
        continue
    depth = re.search(r"Inner Loop Header: Depth=(\d+)", hdr).group(1)
    end = next((e for e in range(k + 1, len(body)) if re.search(r"s_cbranch_\w+ " + re.escape(lab) + r"\b", body[e])), None)
    if end is None:
        continue
    seg = body[k:end + 1]
    valu = [l for l in seg if re.match(r"\s+v_", l)]
    salu = [l for l in seg if re.match(r"\s+s_", l)]
    print(f"{lab} depth {depth}: lines {k}-{end}, VALU {len(valu)} (v_mov {sum('v_mov' in l for l in valu)}, readlane/writelane "
          f"{sum('lane' in l for l in valu)}, rcp {sum('v_rcp' in l for l in valu)}), SALU {len(salu)}, "
          f"labels inside {sum(l.startswith('.LBB') for l in seg) - 1}")
