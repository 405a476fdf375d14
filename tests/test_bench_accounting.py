"""CPU: the accounting bench.py reports (SURVEY.md §8d conventions) — algorithmic bytes, FLOP convention,
CPU share detection, PMC traffic pickup — and that the N>1 launch refuses to run without a launcher."""
import importlib.util
import json
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
spec = importlib.util.spec_from_file_location("bench_module", REPO / "bench.py")
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_algorithmic_bytes_match_survey_table():
    # L mode bytes/eval = 8 D + 8 + 28 Ne / B  ->  cfg2 57.4, cfg3 160.3, cfg4 163.4, cfg5 267.4
    for D, Ne, B, want in ((6, 200, 4096, 57.4), (19, 200, 16384, 160.3), (19, 1000, 8192, 163.4), (32, 2000, 16384, 267.4)):
        assert abs(bench.algorithmic_bytes_per_launch(D, Ne, B) / B - want) < 0.06


def test_flop_convention():
    # F_eval = Ne (Np (128 + 96 n_it) + 60); cfg3 with n_it = 2.9 -> ~256 kFLOP
    assert abs(bench.flops_per_eval(3, 200, 2.9) - 256_000) < 3_000
    assert bench.flops_per_eval(0, 50, 1.0) == 50 * 60


def test_cpu_share_and_overrides(monkeypatch):
    n = bench.host_cpu_share()
    assert 1 <= n <= (len(__import__("os").sched_getaffinity(0)))
    monkeypatch.setenv("RVLL_CPU_THREADS", "3")
    assert bench.host_cpu_share() == 3


def test_pmc_traffic_pickup_applies_the_gfx950_read_correction():
    rec = json.loads((REPO / "profiles" / "pmc_traffic.json").read_text())
    got = bench.pmc_traffic(rec["cfg"], rec["batch"])
    assert got == (2.0 * rec["fetch_kib"] + rec["write_kib"]) * 1024.0       # FETCH_SIZE counts half of a read stream
    assert bench.pmc_traffic(rec["cfg"], rec["batch"] + 1) is None            # only for the measured configuration


def test_multi_gpu_needs_a_launcher():
    out = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env={"PATH": "/usr/bin:/bin", "HOME": "/tmp"})
    assert out.returncode != 0 and "torch.distributed.run" in (out.stderr + out.stdout)
