"""CPU: the oracle's C source under AddressSanitizer + UBSan (sanitizers run on the CPU build
only — GPU ASan is not available on this pool)."""
import os
import shutil
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    exe = tmp_path / "oracle_asan"
    cmd = ["gcc", "-O1", "-g", "-std=gnu11", "-ffp-contract=off", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", f"-I{REPO / 'include'}",
           str(REPO / "tests" / "native" / "oracle_asan_main.c"), str(REPO / "oracle" / "rvll_oracle.c"),
           "-o", str(exe), "-lm"]
    subprocess.run(cmd, check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "oracle sanitizer run ok" in out.stdout and "ERROR" not in out.stderr
