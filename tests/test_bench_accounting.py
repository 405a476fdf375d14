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


def test_pmc_traffic_pickup_applies_the_gfx950_read_correction(tmp_path, monkeypatch):
    rec = json.loads((REPO / "profiles" / "pmc_traffic.json").read_text())
    got, source = bench.pmc_traffic(rec["cfg"], rec["batch"])
    assert got == (2.0 * rec["fetch_kib"] + rec["write_kib"]) * 1024.0       # FETCH_SIZE counts half of a read stream
    assert "profiles/pmc_traffic.json" in source
    assert bench.pmc_traffic(rec["cfg"], rec["batch"] + 1)[0] is None          # only for the measured configuration


def test_pmc_traffic_record_is_stamped_with_the_kernel_sources_it_was_measured_with(monkeypatch):
    """roofline.traffic is copied from a committed rocprofv3 PMC record, so the record carries a hash of the kernel
    sources; bench.py drops the number (traffic: null, traffic_source says why) once the sources have changed, and
    this test keeps the committed record current."""
    rec = json.loads((REPO / "profiles" / "pmc_traffic.json").read_text())
    assert rec["kernel_source_sha"] == bench.kernel_source_sha(), \
        "kernel sources changed: re-run scripts/profile_gpu.sh on the GPU box and commit its pmc_traffic.json"
    monkeypatch.setattr(bench, "kernel_source_sha", lambda: "0" * 16)
    got, why = bench.pmc_traffic(rec["cfg"], rec["batch"])
    assert got is None and "stale" in why
