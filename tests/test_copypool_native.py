"""CPU: the worker-thread pool behind the streamed host batches (evidence_amd/csrc/rvll_copypool.h) under ThreadSanitizer
and under AddressSanitizer + UBSan — the GPU box has no sanitizers, and the pool is plain C++."""
import os
import shutil
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]
SRC = REPO / "tests" / "native" / "copypool_main.cpp"


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_copy_pool_is_clean_under_sanitizers(tmp_path, sanitizer):
    exe = tmp_path / "copypool"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-pthread", f"-fsanitize={sanitizer}", "-fno-omit-frame-pointer",
           f"-I{REPO / 'evidence_amd' / 'csrc'}", str(SRC), "-o", str(exe)]
    subprocess.run(cmd, check=True)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "copy pool run ok" in out.stdout and "WARNING: ThreadSanitizer" not in out.stderr and "ERROR" not in out.stderr
