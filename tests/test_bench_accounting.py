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


WATCHDOG_CHILD = """
import importlib.util, json, sys, time
spec = importlib.util.spec_from_file_location("bench_module", {bench!r})
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
wd = bench.Watchdog(0, 0.3)
wd.hung_line = {{"metric": "m", "value": 0.0, "config": {{"workload": "w"}}}}
if {with_fallback}:
    wd.fallback_line = {{"metric": "m", "value": 5.0, "config": {{"workload": "w", "allgather": "host-socket-fallback"}}}}
wd.kick("communicator")
time.sleep(30)                                   # a collective that never returns
"""


def test_watchdog_reports_the_last_completed_measurement_and_leaves():
    # a phase that hangs after a measurement completed: that line goes out (with the phase named), exit code 0
    code = WATCHDOG_CHILD.format(bench=str(REPO / "bench.py"), with_fallback=True)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] == 5.0 and "communicator" in line["config"]["note"]
    assert "no progress" in r.stderr
    # nothing completed: the line says so (value 0, rccl-hung) and the exit code is non-zero
    code = WATCHDOG_CHILD.format(bench=str(REPO / "bench.py"), with_fallback=False)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] == 0.0 and line["config"]["allgather"] == "rccl-hung" and line["config"]["hung_phase"] == "communicator"
