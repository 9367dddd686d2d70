"""GPU parity: point-cloud kernels (C ABI) vs the CPU oracle and the reference's golden clouds."""
import ctypes

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from ssrlcv_amd import capi
    return capi


def _bundles_gpu(capi, v, stage):
    mm, kp, cams = v["mm%d" % stage], v["kp%d" % stage], v["cameras"]
    mm_d, kp_d, cam_d = capi.to_dev(mm), capi.to_dev(kp), capi.to_dev(cams)
    b_d, l_d = capi.generate_bundles(mm_d, kp_d, len(mm), cam_d, len(cams), len(kp))
    return b_d, l_d


@pytest.mark.parametrize("view,nview", [("Pipeline2View", False), ("Pipeline3View", True)])
def test_bundles_and_triangulation_match_oracle_and_fixture(capi, oracle_lib, view, nview):
    v = H.load_view(view)
    n = len(v["mm0"])
    b_d, l_d = _bundles_gpu(capi, v, 0)
    ob, ol, _ = H.oracle_bundles(oracle_lib, v["mm0"], v["kp0"], v["cameras"])
    gb = capi.to_host(b_d, H.BUNDLE, n)
    gl = capi.to_host(l_d, H.LINE, len(v["kp0"]))
    assert np.array_equal(gb["numLines"], ob["numLines"]) and np.array_equal(gb["index"], ob["index"])
    assert (gb["invalid"] == 0).all()
    assert np.array_equal(gl["pnt"], ol["pnt"])
    # direction vectors: tanf / sinf / cosf are the shared sv_math.h functions on both sides -> bit-equal
    assert np.array_equal(H.bits(gl["vec"]), H.bits(ol["vec"]))
    # ... and so the clouds are too (pure +-*/ and sqrt after that)
    pts0_d, _, _ = capi.triangulate(l_d, b_d, n, nview=nview)
    assert np.array_equal(pts0_d.cpu().numpy().reshape(-1, 3), H.oracle_triangulate(oracle_lib, nview, ob, ol)[0])
    pts_d, err_d, sum_d = capi.triangulate(l_d, b_d, n, nview=nview, want_errors=True)
    pts = pts_d.cpu().numpy().reshape(-1, 3)
    opts, oerr, osum = H.oracle_triangulate(oracle_lib, nview, ob, ol, want_errors=True)
    # fed with the ORACLE's lines the kernel must reproduce the oracle bit-for-bit (pure +-*/ and sqrt)
    pts2_d, err2_d, sum2_d = capi.triangulate(capi.to_dev(ol), capi.to_dev(ob), n, nview=nview, want_errors=True)
    assert np.array_equal(pts2_d.cpu().numpy().reshape(-1, 3), opts)
    assert np.array_equal(err2_d.cpu().numpy(), oerr)
    assert abs(float(sum2_d.item()) - osum) <= 1e-4 * max(1.0, abs(osum))  # float sum order differs (atomics)
    # end to end against the reference's own cloud: every point bit for bit (13 534 two-view, 21 177 N-view)
    assert np.array_equal(pts.view(np.uint32), v["points0"].view(np.uint32))


@pytest.mark.parametrize("view,nview", [("Pipeline2View", False), ("Pipeline3View", True)])
def test_filtered_clouds_are_bit_equal_to_the_fixtures(capi, view, nview):
    """1_KeyPoint / 1_MultiMatch (the match sets after upstream's filters) -> 1_6float3 (= 2_6float3 for two views)."""
    v = H.load_view(view)
    b_d, l_d = _bundles_gpu(capi, v, 1)
    pts_d, _, _ = capi.triangulate(l_d, b_d, len(v["mm1"]), nview=nview)
    assert np.array_equal(pts_d.cpu().numpy().reshape(-1, 3).view(np.uint32), v["points1"].view(np.uint32))


def test_cutoff_and_void_variants(capi, oracle_lib):
    v = H.load_view("Pipeline2View")
    n = len(v["mm0"])
    ob, ol, _ = H.oracle_bundles(oracle_lib, v["mm0"], v["kp0"], v["cameras"])
    _, oerr, _ = H.oracle_triangulate(oracle_lib, False, ob.copy(), ol, want_errors=True)
    cutoff = float(np.median(oerr))
    ob2 = ob.copy()
    H.oracle_triangulate(oracle_lib, False, ob2, ol, want_errors=True, cutoff=cutoff)
    b_d = capi.to_dev(ob)
    # void variant: no point cloud, only invalid flags + error sum
    _, _, sum_d = capi.triangulate(capi.to_dev(ol), b_d, n, want_points=False, cutoff=cutoff)
    gb = capi.to_host(b_d, H.BUNDLE, n)
    assert np.array_equal(gb["invalid"], ob2["invalid"])
    assert 0 < gb["invalid"].sum() < n
    assert np.isfinite(sum_d.item())


def test_nview_no_error_variant_flags_singular_bundles(capi):
    # two identical lines: S = sum(vv^T - I) is singular -> invalid (src/PointCloudFactory.cu:4923-4926)
    lines = np.zeros(4, H.LINE)
    lines["vec"][:] = [[0, 0, 1], [0, 0, 1], [1, 0, 0], [0, 1, 0]]
    lines["pnt"][:] = [[0, 0, 0], [1, 0, 0], [0, 0, 0], [0, 0, 0]]
    bundles = np.zeros(2, H.BUNDLE)
    bundles["numLines"] = 2
    bundles["index"] = [0, 2]
    b_d = capi.to_dev(bundles)
    capi.triangulate(capi.to_dev(lines), b_d, 2, nview=True, no_error_variant=True)
    gb = capi.to_host(b_d, H.BUNDLE, 2)
    assert list(gb["invalid"]) == [1, 0]


def test_ba_sweep_matches_oracle(capi, oracle_lib):
    """612 camera-parameter sets as calculateImageGradient/Hessian build them: one fused launch vs 612 oracle evals."""
    v = H.load_view("Pipeline2View")
    mm, kp, cams = v["mm1"], v["kp1"], v["cameras"]
    base = np.concatenate([np.concatenate([c["cam_pos"], c["cam_rot"]]) for c in cams]).astype(np.float32)
    rng = np.random.default_rng(5)
    K = 612
    params = np.tile(base, (K, 1))
    for k in range(1, K):
        i, j = rng.integers(0, 12, 2)
        params[k, i] += np.float32(1e-4 if i % 6 < 3 else 1e-5)
        params[k, j] -= np.float32(1e-4 if j % 6 < 3 else 1e-5)
    sums = capi.ba_sweep2(capi.to_dev(mm), capi.to_dev(kp), len(mm), capi.to_dev(cams), len(cams),
                          capi.to_dev(params), K).cpu().numpy()
    oracle_lib.oracle_ba_eval.restype = ctypes.c_float
    for k in (0, 1, 17, 300, 611):
        ref = oracle_lib.oracle_ba_eval(ctypes.c_uint32(len(mm)), H.P(mm), H.P(kp), H.P(cams), ctypes.c_uint32(2),
                                        H.P(np.ascontiguousarray(params[k])))
        assert abs(sums[k] - ref) <= 2e-3 * max(1.0, abs(ref)), (k, sums[k], ref)
    assert np.isfinite(sums).all()


def test_pushbroom_bundles_match_oracle(capi, oracle_lib):
    rng = np.random.default_rng(9)
    ncam, nb = 8, 5000
    pb = np.zeros(ncam, H.PUSHBROOM)
    pb["axis_radius"] = 6371.0
    pb["roll"] = rng.uniform(-8, 8, ncam)
    pb["altitude"] = 400.0
    pb["foc"] = 0.859311
    pb["fov"] = 0.0418879
    pb["gsd"] = 0.006
    pb["dpix"] = 3.5e-5
    pb["size"] = 8192
    mm = np.zeros(nb, H.MULTIMATCH)
    mm["numKeyPoints"] = rng.integers(2, 5, nb)
    mm["index"] = np.concatenate([[0], np.cumsum(mm["numKeyPoints"])[:-1]])
    nk = int(mm["numKeyPoints"].sum())
    kp = np.zeros(nk, H.KEYPOINT)
    kp["parentId"] = rng.integers(0, ncam, nk)
    kp["loc"] = rng.uniform(0, 8192, (nk, 2))
    b_d, l_d = capi.generate_pushbroom_bundles(capi.to_dev(mm), capi.to_dev(kp), nb, capi.to_dev(pb), ncam, nk)
    ob = np.zeros(nb, H.BUNDLE)
    ol = np.zeros(nk, H.LINE)
    oracle_lib.oracle_generate_pushbroom_bundles(ctypes.c_uint32(nb), H.P(mm), H.P(kp), H.P(pb), H.P(ob), H.P(ol))
    gl = capi.to_host(l_d, H.LINE, nk)
    # shared tanf / sinf / cosf: bit-equal (round 1: two libms, 1e-5 / 1e-3)
    assert np.array_equal(H.bits(gl["vec"]), H.bits(ol["vec"]))
    assert np.array_equal(H.bits(gl["pnt"]), H.bits(ol["pnt"]))
    assert np.array_equal(capi.to_host(b_d, H.BUNDLE, nb)["index"], mm["index"])


# ---- filters between triangulation and BA on the device (csrc/filter.hip) ------------------------------------------------
@pytest.mark.parametrize("n,jump", [(1, 10), (9, 10), (10, 10), (1023, 10), (13534, 10), (21177, 10), (100000, 3), (1 << 20, 10),
                                    (3100000, 10), (5000, 1)])
def test_sample_cutoff_is_the_hosts_sequential_sum(capi, oracle_lib, n, jump):
    """sigma x std of every jump-th error: the device chain must be the host loop's float sums bit for bit
    (src/PointCloudFactory.cu:3121-3156)."""
    import torch
    rng = np.random.default_rng(n + jump)
    errs = (rng.random(n).astype(np.float32) ** 4 * np.float32(50.0)).astype(np.float32)   # skewed, like squared line gaps
    oracle_lib.oracle_sample_cutoff.restype = ctypes.c_float
    want = np.float32(oracle_lib.oracle_sample_cutoff(H.P(errs), ctypes.c_uint32(n), ctypes.c_uint32(jump), ctypes.c_float(3.0)))
    got = capi.error_sample_cutoff(torch.from_numpy(errs).cuda(), n, jump, 3.0).cpu().numpy()[0]
    if n < jump:
        assert np.isnan(got) and np.isnan(want)   # an empty sample: 0 / 0 on both sides
    else:
        assert got.view(np.uint32) == want.view(np.uint32), (got, want)


@pytest.mark.parametrize("n,maxlines,pbad", [(1, 2, 0.0), (5, 2, 1.0), (1000, 2, 0.3), (1024, 5, 0.5), (70000, 2, 0.02), (300000, 8, 0.66),
                                             (1 << 20, 3, 0.001)])
def test_filter_matchset_equals_the_host_rebuild(capi, oracle_lib, n, maxlines, pbad):
    """ssrlcv_hip_filter_matchset (one pass, three running sums by decoupled look-back) against the reference's host loop
    (src/PointCloudFactory.cu:3253-3268) on random flag patterns and bundle sizes, from one tile to hundreds."""
    rng = np.random.default_rng(n)
    bundles = np.zeros(n, H.BUNDLE)
    bundles["numLines"] = rng.integers(2, maxlines + 1, n)
    bundles["index"] = np.concatenate([[0], np.cumsum(bundles["numLines"])[:-1]])
    bundles["invalid"] = rng.random(n) < pbad
    nk = int(bundles["numLines"].sum())
    kp = np.zeros(nk, H.KEYPOINT)
    kp["parentId"] = rng.integers(0, 8, nk)
    kp["loc"] = rng.random((nk, 2)).astype(np.float32) * 4096
    mm_o, kp_o, cnt_o = np.zeros(n, H.MULTIMATCH), np.zeros(nk, H.KEYPOINT), np.zeros(3, np.uint32)
    oracle_lib.oracle_filter_matchset(ctypes.c_uint32(n), H.P(bundles), H.P(kp), H.P(mm_o), H.P(kp_o), H.P(cnt_o))
    mm_d, kp_d, counts = capi.filter_matchset(capi.to_dev(bundles), capi.to_dev(kp), n, nk)
    got = counts.cpu().numpy().astype(np.uint32)
    assert np.array_equal(got, cnt_o), (got, cnt_o)
    mm_g = capi.to_host(mm_d, H.MULTIMATCH, int(got[0]))
    kp_g = capi.to_host(kp_d, H.KEYPOINT, int(got[1]))
    assert np.array_equal(mm_g["numKeyPoints"], mm_o["numKeyPoints"][:got[0]]) and np.array_equal(mm_g["index"], mm_o["index"][:got[0]])
    assert np.array_equal(kp_g["parentId"], kp_o["parentId"][:got[1]]) and np.array_equal(kp_g["loc"], kp_o["loc"][:got[1]])


@pytest.mark.parametrize("view,after", [("Pipeline2View", 13308), ("Pipeline3View", 21099)])
def test_device_filters_reproduce_the_stage1_fixtures(capi, oracle_lib, view, after):
    """doFiltering's sequence (src/Pipeline.cu:305-340) through pipeline.apply_filters on the device copies of the
    reference's stage-0 MatchSet -> the reference's stage-1 MatchSet, every key point and index; equal to the oracle's
    filters step by step."""
    from ssrlcv_amd import pipeline
    v = H.load_view(view)
    mm, kp, cams = v["mm0"], v["kp0"], v["cameras"]
    dev = {}
    mm_f, kp_f = pipeline.apply_filters(mm, kp, dev, cams, pipeline.REFERENCE_FILTERS)
    assert len(mm_f) == after == len(v["mm1"])
    assert np.array_equal(mm_f["numKeyPoints"], v["mm1"]["numKeyPoints"]) and np.array_equal(mm_f["index"], v["mm1"]["index"])
    assert np.array_equal(kp_f["parentId"], v["kp1"]["parentId"]) and np.array_equal(kp_f["loc"], v["kp1"]["loc"])
    # the device copies left for the triangulation are the same arrays
    assert np.array_equal(capi.to_host(dev["matches"], H.MULTIMATCH, after)["index"], v["mm1"]["index"])
    # and the filtered cloud is the reference's stage-1 cloud, bit for bit
    b_d, l_d = capi.generate_bundles(dev["matches"], dev["keypoints"], after, capi.to_dev(cams), len(cams), len(kp_f))
    pts_d, _, _ = capi.triangulate(l_d, b_d, after, nview=len(cams) > 2)
    assert np.array_equal(pts_d.cpu().numpy().reshape(-1, 3).view(np.uint32), v["points1"].view(np.uint32))
    # a single statistical step against the oracle's
    mo, ko = H.oracle_filter(oracle_lib, mm, kp, cams, "statistical", sigma=2.0, sample_size=0.25)
    mg, kg = pipeline.apply_filters(mm, kp, {}, cams, [("statistical", 2.0, 0.25)])
    assert len(mo) < len(mm) and np.array_equal(mg["index"], mo["index"]) and np.array_equal(kg["loc"], ko["loc"])
