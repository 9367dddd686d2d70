"""The pooled host arrays behind pipeline.build_match_set's results: a buffer is handed out again only
when no result array, slice or view of it is alive -- judged by a reference count calibrated on this interpreter."""
import numpy as np


def test_pooled_result_buffers_are_never_shared_with_live_results():
    from ssrlcv_amd import pipeline as p
    assert p._POOL_FREE_COUNT is not None, "the calibration must succeed on the interpreter the suite runs on"
    slot = "test-pool"
    first = p._result_buffer(1000, slot)
    ident = id(first)
    held = first[:800].view(np.uint16)[10:20]      # what a caller keeps: a slice of a view of the result
    del first
    other = p._result_buffer(1000, slot)           # `held` still refers to the first buffer: a different one comes back
    assert id(other) != ident
    other_id = id(other)
    del other
    del held
    again = p._result_buffer(1000, slot)           # now both are free: the first one in the pool is reused
    assert id(again) in (ident, other_id)
    del again
    big = p._result_buffer(1 << 20, slot)          # too small ones are skipped
    assert big.size >= 1 << 20


def test_pool_is_bypassed_when_the_count_cannot_be_trusted(monkeypatch):
    from ssrlcv_amd import pipeline as p
    monkeypatch.setattr(p, "_POOL_FREE_COUNT", None)
    a = p._result_buffer(100, "test-pool-off")
    ida = id(a)
    b = p._result_buffer(100, "test-pool-off")
    assert id(b) != ida
