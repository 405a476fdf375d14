#!/usr/bin/env python3
"""Tile form vs CU-wide form of the log-L kernel over batch sizes (HIP-event kernel time per launch, back-to-back
launches on one stream): where does the CU-wide form start to win?  Feeds choose_cu_form() in rvll_api.hip.

    python scripts/form_sweep.py > profiles/rNN_form_sweep.txt
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from evidence_amd import GpuRVModel  # noqa: E402
from evidence_amd.synthetic import make_workload  # noqa: E402


def main():
    sizes = {1: [400, 4096, 32768, 131072], 2: [1024, 2048, 4096, 8192, 16384, 65536],
             3: [1024, 2048, 4096, 6144, 8192, 12288, 16384, 20000, 32768, 65536, 262144],
             4: [256, 512, 1024, 2048, 4096, 8192, 65536], 5: [256, 512, 1024, 2048, 4096, 16384]}
    print("cfg  epochs  batch    wave rounds/CU   tile us (PB)      cu us (PB x tiles)    cu/tile   auto picks")
    for cfg, bs in sizes.items():
        w = make_workload(cfg)
        with GpuRVModel(w.fixedpardict, w.table, w.parnames) as m:
            for b in bs:
                theta = w.sample_theta(b, seed=7)
                m.dev_upload_theta(theta)
                res = {}
                for form in ("tile", "cu", "auto"):
                    m.set_kernel_form(form)
                    res[form] = m.dev_time_loglike(b, warmup=20, iters=100)
                t, c, a = res["tile"], res["cu"], res["auto"]
                rounds = b * w.table.n_epochs / 64 / 256
                cu_txt = (f"{c['kernel_ms_median'] * 1e3:8.2f} ({c['points_per_block']:3d} x {c['blocks']:5d})"
                          if c["threads"] == 1024 else "   (does not fit)     ")
                print(f"{cfg:3d} {w.table.n_epochs:7d} {b:7d} {rounds:12.1f}    {t['kernel_ms_median'] * 1e3:8.2f} ({t['points_per_block']:2d})    "
                      f"{cu_txt}   {c['kernel_ms_median'] / t['kernel_ms_median']:6.3f}    {'cu' if a['threads'] == 1024 else 'tile'}")


if __name__ == "__main__":
    main()
