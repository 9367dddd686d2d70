"""Round-5 entry points through the C ABI on the GPU: the pinned-bounce form of ssrlcv_hip_memcpy for pageable host memory,
the pair-bundle selection pass, and the stage-boundary event of the fused extract."""
import ctypes

import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nbytes", [(4 << 20) - 1, 4 << 20, (8 << 20) + 5, (20 << 20) + 3, (33 << 20) + 4097])
def test_memcpy_to_and_from_pageable_memory_is_exact(capi, nbytes):
    """ssrlcv_hip_memcpy (cudaMemcpy's stand-in, include/Unity.cuh:820-854): copies of 4 MB and more to / from memory the
    runtime does not know as pinned go through two pinned 8 MB bounce buffers and a small team of copy threads
    (csrc/capi_common.hip); below that, and for pinned memory, plain hipMemcpy.  Sizes around the threshold and the chunk
    size, odd tails, both directions, twice (the bounce buffers are reused): every byte arrives."""
    rng = np.random.default_rng(nbytes & 0xffff)
    for rep in range(2):
        src = rng.integers(0, 256, nbytes, dtype=np.uint8)   # plain numpy memory: pageable
        dev = capi.dev_bytes(nbytes + 64)
        dev.zero_()
        assert capi.LIB.ssrlcv_hip_memcpy(ctypes.c_void_p(dev.data_ptr() + 16), H.P(src), ctypes.c_size_t(nbytes), ctypes.c_int(0)) == 0
        torch.cuda.synchronize()
        got = dev.cpu().numpy()
        assert np.array_equal(got[16:16 + nbytes], src) and not got[:16].any() and not got[16 + nbytes:].any()
        back = np.zeros(nbytes + 32, np.uint8)
        assert capi.LIB.ssrlcv_hip_memcpy(ctypes.c_void_p(back.ctypes.data + 8), ctypes.c_void_p(dev.data_ptr() + 16), ctypes.c_size_t(nbytes),
                                          ctypes.c_int(1)) == 0
        assert np.array_equal(back[8:8 + nbytes], src) and not back[:8].any() and not back[8 + nbytes:].any()
    # a copy that follows work on the null stream sees its result (the staged path synchronises the null stream first)
    t = torch.zeros(6 << 20, dtype=torch.uint8, device="cuda")
    with torch.cuda.stream(torch.cuda.default_stream()):
        t.fill_(7)
        out = np.zeros(6 << 20, np.uint8)
        assert capi.LIB.ssrlcv_hip_memcpy(H.P(out), ctypes.c_void_p(t.data_ptr()), ctypes.c_size_t(out.size), ctypes.c_int(1)) == 0
    assert (out == 7).all()
    assert capi.LIB.ssrlcv_hip_memcpy(None, H.P(out), ctypes.c_size_t(16), ctypes.c_int(0)) != 0   # null pointers are refused


@pytest.mark.parametrize("pair", [(0, 1), (0, 2), (1, 2), (2, 1)])
def test_select_pair_bundles_equals_the_host_selection(capi, pair):
    """ssrlcv_hip_select_pair_bundles on the reference's golden 3-view MatchSet (21 177 multi-matches, 51 442 key points):
    the two-view bundles of an image pair as a two-camera MatchSet, against the selection written out in numpy."""
    v = H.load_view("Pipeline3View")
    mm, kp = v["mm0"], v["kp0"]
    mm_d, kp_d, cnt = capi.select_pair_bundles(capi.to_dev(mm), capi.to_dev(kp), len(mm), len(kp), pair[0], pair[1])
    n = int(cnt.item())
    idx = mm["index"]
    two = (mm["numKeyPoints"] == 2) & (kp["parentId"][idx] == pair[0]) & (kp["parentId"][np.minimum(idx + 1, len(kp) - 1)] == pair[1])
    rows = np.nonzero(two)[0]
    # (only image 0 seeds multi-matches in a 3-view set, src/MatchFactory.cu:969: the pairs that do not start there are empty)
    assert n == len(rows) and (n > 1000 or pair[0] != 0)
    got_mm = capi.to_host(mm_d, H.MULTIMATCH, n)
    got_kp = capi.to_host(kp_d, H.KEYPOINT, 2 * n)
    assert (got_mm["numKeyPoints"] == 2).all() and np.array_equal(got_mm["index"], 2 * np.arange(n))
    assert np.array_equal(got_kp["loc"][0::2], kp["loc"][idx[rows]]) and np.array_equal(got_kp["loc"][1::2], kp["loc"][idx[rows] + 1])
    assert (got_kp["parentId"][0::2] == 0).all() and (got_kp["parentId"][1::2] == 1).all() and not got_kp["pad"].any()
    # empty input: the count is zeroed, nothing else is touched
    _, _, c0 = capi.select_pair_bundles(capi.dev_bytes(8), capi.dev_bytes(16), 0, 0, 0, 1)
    assert int(c0.item()) == 0


def test_stage_event_splits_the_fused_extract(capi, oracle_lib):
    """ssrlcv_sift_plan_set_stage_event: the event is recorded between the two stages of every later extract on the plan (the
    hook bench.py times the stages with), the features are those of the two stand-alone stage calls, and clearing it stops
    the recording."""
    img = H.synthetic_image(512, 384, seed=9)
    pix = capi.to_dev(img)
    plan = capi.SiftPlan(512, 384)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    plan.set_stage_event(e1)
    e0.record()
    plan.extract(pix)
    e2.record()
    torch.cuda.synchronize()
    fused = plan.features_host(H.FEATURE)
    a, b = e0.elapsed_time(e1), e1.elapsed_time(e2)
    assert a > 0.0 and b > 0.0, (a, b)
    plan.set_stage_event(None)
    plan.build_dog(pix)
    plan.describe()
    split = plan.features_host(H.FEATURE)
    H.assert_features_equal(fused, split)
    H.assert_features_equal(fused, H.oracle_sift(oracle_lib, img))
