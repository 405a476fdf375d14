#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel in evidence_amd/csrc (hipcc -Rpass-analysis=kernel-resource-usage).

    python scripts/kernel_resources.py [> profiles/rNN_kernel_resources.txt]
"""
import re
import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent.parent / "evidence_amd" / "csrc"
FLAGS = "-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -fvisibility=hidden -I../../include -I."


def main():
    rows = []
    for src, extra in (("rvll_kernels.hip", ""), ("rvll_walk.hip", "-mllvm -disable-machine-licm"), ("rvll_live.hip", ""), ("rvll_fip.hip", "")):
        cmd = f"/opt/rocm/bin/hipcc {FLAGS} {extra} -Rpass-analysis=kernel-resource-usage -c {src} -o /dev/null"
        err = subprocess.run(cmd, shell=True, cwd=CSRC, capture_output=True, text=True).stderr
        cur = None
        for line in err.splitlines():
            m = re.search(r"remark:\s*(.*?)\s*\[-Rpass", line)
            if not m:
                continue
            text = m.group(1).strip()
            if text.startswith("Function Name:"):
                name = text.split(":", 1)[1].strip()
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                cur = {"name": re.sub(r"rvll::\(anonymous namespace\)::", "", dem).split("(")[0]}
                rows.append(cur)
            elif cur is not None and ":" in text:
                k, v = text.split(":", 1)
                cur[k.strip()] = v.strip()
    cols = ["VGPRs", "AGPRs", "TotalSGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"]
    print(f"{'kernel':58s} " + " ".join(f"{c.split(' [')[0][:10]:>10s}" for c in cols))
    for r in rows:
        print(f"{r['name'][:58]:58s} " + " ".join(f"{r.get(c, '-'):>10s}" for c in cols))
    return 0


if __name__ == "__main__":
    sys.exit(main())
