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
