import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        from evidence_amd import _abi
        import ctypes as C
        n = C.c_int32(0)
        return _abi.load().rvll_device_count(C.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_required():
    """GPU tests must FAIL (not skip) on a GPU box whose HIP library is missing; they are only
    deselected through `-m "not gpu"`."""
    from evidence_amd import _abi
    _abi.load()
    assert _gpu_available(), "no HIP device visible: -m gpu tests need the MI355X"
