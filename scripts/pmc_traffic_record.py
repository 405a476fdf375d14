#!/usr/bin/env python3
"""Write profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/profile_gpu.sh, stamped with the
hash of the kernel sources (bench.py kernel_source_sha) so that a stale record is recognised.

    python scripts/pmc_traffic_record.py gpurun_out/<tag> [cfg batch]  > profiles/pmc_traffic.json
"""
import importlib.util
import json
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "scripts"))
from pmc_summary import summarise  # noqa: E402

spec = importlib.util.spec_from_file_location("bench_module", REPO / "bench.py")
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def main():
    tag = Path(sys.argv[1])
    cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
    mean, cnt = summarise(str(tag / "pmc*" / "*" / "*_counter_collection.csv"), ("loglike",))
    kernels = sorted({k for k, c in mean if c == "FETCH_SIZE"})
    if len(kernels) != 1:
        sys.exit(f"expected one log-L kernel with FETCH_SIZE rows, found {kernels}")
    k = kernels[0]
    print(json.dumps({
        "cfg": cfg, "batch": batch, "kernel": k, "fetch_kib": mean[(k, "FETCH_SIZE")], "write_kib": mean[(k, "WRITE_SIZE")],
        "kernel_source_sha": bench.kernel_source_sha(),
        "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, mean over {cnt[(k, 'FETCH_SIZE')]} dispatches of "
                  "`bench.py --steps 50 --warmup 5 --no-cpu --no-extras` (scripts/profile_gpu.sh)",
        "note": "FETCH_SIZE is doubled by bench.py per the gfx950 half-count of coalesced read streams (MI355X_MICROARCH.md, HBM); "
                "theta is re-read from the Infinity Cache between back-to-back launches; WRITE_SIZE = log-L + flags",
    }, indent=1))


if __name__ == "__main__":
    main()
