"""Large result arrays of the host-buffer calls come from a small store of recycled blocks (evidence_amd/engine.py,
_result_array): a block goes back only when the last array looking at it is gone."""
import gc

import numpy as np
import pytest

from evidence_amd import engine


@pytest.fixture(autouse=True)
def _empty_store():
    engine._result_blocks.clear()
    yield
    engine._result_blocks.clear()


def _addr(a):
    return a.__array_interface__["data"][0]


def test_small_results_are_plain_arrays():
    a = engine._result_array((1000, 19))
    assert a.flags.owndata and a.shape == (1000, 19) and a.dtype == np.float64


def test_large_result_is_an_ordinary_writable_float64_array():
    rows = engine._RESULT_MIN_BYTES // (8 * 19) + 1
    a = engine._result_array((rows, 19))
    assert a.shape == (rows, 19) and a.dtype == np.float64 and a.flags.c_contiguous and a.flags.writeable
    a[:] = 2.0
    assert float(a.sum()) == 2.0 * rows * 19
    b = a.copy()
    assert b.flags.owndata and np.array_equal(a, b)


def test_block_returns_only_when_the_last_view_is_gone_and_is_reused():
    rows = engine._RESULT_MIN_BYTES // (8 * 19) + 1
    a = engine._result_array((rows, 19))
    where = _addr(a)
    tail = a[-3:]                       # a view keeps the memory alive after the result itself is dropped
    tail[:] = 7.0
    del a
    gc.collect()
    assert engine._result_blocks == []
    other = engine._result_array((rows, 19))      # while the view lives, a new result must be other memory
    assert _addr(other) != where
    other[:] = 1.0
    assert np.all(tail == 7.0)
    del tail
    gc.collect()
    assert len(engine._result_blocks) == 1
    again = engine._result_array((rows, 19))
    assert _addr(again) == where and engine._result_blocks == []


def test_store_is_bounded():
    rows = engine._RESULT_MIN_BYTES // 8 + 1
    held = [engine._result_array((rows + i,)) for i in range(engine._RESULT_KEEP + 2)]
    del held
    gc.collect()
    assert len(engine._result_blocks) == engine._RESULT_KEEP
