"""GPU parity: the MFMA 128-D matcher (C ABI; int8 formulation by default, the fp16 one in tests/test_gpu_configs.py) vs
the CPU oracle -- indices and distances bit-exact."""
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

REL, ABS = 0.6, 200.0 * 200.0


@pytest.fixture(scope="module")
def capi():
    from ssrlcv_amd import capi
    return capi


def random_features(n, seed, lo=0, hi=256, w=1024):
    rng = np.random.default_rng(seed)
    f = np.zeros(n, H.FEATURE)
    f["parent"] = -1
    f["values"] = rng.integers(lo, hi, (n, 128), dtype=np.uint8) if n else 0
    f["loc"] = rng.uniform(0, w, (n, 2)).astype(np.float32)
    f["sigma"] = 1.0
    return f


def run_gpu(capi, mode, q, t, kind, seed=None, eps=0.0, delta=0.0, cam=None, proj=None, rel=REL, absolute=ABS,
            qid=3, tid=7):
    params = capi.make_match_params(mode, qid, tid, eps, delta, rel, absolute, cam, proj)
    out_d = capi.match(capi.to_dev(q), len(q), capi.to_dev(t) if len(t) else None, len(t), params, kind,
                       seed_d=capi.to_dev(seed) if seed is not None else None)
    dt = {capi.OUT_DMATCH: H.DMATCH, capi.OUT_UINT2_PAIR: H.UINT2_PAIR, capi.OUT_MATCH: H.MATCH}[kind]
    return capi.to_host(out_d, dt, len(q))


def assert_dmatch_equal(g, o):
    assert np.array_equal(g["invalid"], o["invalid"])
    assert np.array_equal(g["distance"], o["distance"])
    ok = o["invalid"] == 0
    for name in ("kp0_parent", "kp1_parent"):
        assert np.array_equal(g[name][ok], o[name][ok]), name
    for name in ("kp0_loc", "kp1_loc"):  # bit patterns: locations may hold NaN in the degenerate-geometry test
        assert np.array_equal(g[name][ok].view(np.uint32), o[name][ok].view(np.uint32)), name


@pytest.mark.parametrize("nq,nt", [(1, 1), (37, 5), (513, 1000), (2048, 4097), (700, 33)])
def test_brute_force_full_range_u8_is_exact(capi, oracle_lib, nq, nt):
    """Uniform 0..255 bytes: distances up to ~2.8M exercise all three norm digits and the 2^24 headroom."""
    q, t = random_features(nq, 1), random_features(nt, 2)
    big = 3.0e7
    g = run_gpu(capi, 0, q, t, capi.OUT_DMATCH, absolute=big)
    o = H.oracle_match_dmatch(oracle_lib, 0, 3, q, 7, t, None, None, 0, 0, None, REL, big)
    assert_dmatch_equal(g, o)
    assert (g["invalid"] == 0).all()


def test_extreme_descriptors_exact(capi, oracle_lib):
    """All-0 vs all-255 rows: the largest possible distance 128*255^2 = 8,323,200 and the largest cross term."""
    q = random_features(64, 3)
    t = random_features(96, 4)
    q["values"][0] = 255
    q["values"][1] = 0
    t["values"][0] = 0
    t["values"][1] = 255
    t["values"][2] = 255
    g = run_gpu(capi, 0, q, t, capi.OUT_DMATCH, absolute=3.0e7)
    o = H.oracle_match_dmatch(oracle_lib, 0, 3, q, 7, t, None, None, 0, 0, None, REL, 3.0e7)
    assert_dmatch_equal(g, o)
    # restrict the targets to the extremes to force the 8,323,200 distance
    g2 = run_gpu(capi, 0, q[:2], t[:1], capi.OUT_DMATCH, absolute=3.0e7)
    assert g2["distance"][0] == 128 * 255 * 255 and g2["distance"][1] == 0


def test_tie_break_is_lowest_lane_then_lowest_index(capi, oracle_lib):
    """Duplicated targets: winner = smallest (distance, f mod 32, f), not smallest f (src/MatchFactory.cu:2256-2271)."""
    q = random_features(300, 5, hi=40)
    t = random_features(1500, 6, hi=40)
    rng = np.random.default_rng(7)
    # plant exact copies of query descriptors at several target indices with different f mod 32
    for qi in range(0, 300, 3):
        for f in rng.choice(1500, 4, replace=False):
            t["values"][f] = q["values"][qi]
    g = run_gpu(capi, 0, q, t, capi.OUT_UINT2_PAIR)
    o = H.oracle_match_pairs(oracle_lib, 0, 3, q, 7, t, None, None, 0, 0, None, REL, ABS)
    assert np.array_equal(g["a"], o["a"]) and np.array_equal(g["b"], o["b"])
    # the planted queries all matched (distance 0) and at least one winner is not the lowest duplicate index
    planted = g["b"][0::3]
    assert (planted[:, 0] == 7).all()


def test_absolute_threshold_and_seed_ratio(capi, oracle_lib):
    q = random_features(1000, 8, hi=64)
    t = random_features(3000, 9, hi=64)
    s = random_features(777, 10, hi=64)
    sd_g = capi.seed_distances(capi.to_dev(q), len(q), capi.to_dev(s), len(s)).cpu().numpy()
    sd_o = H.oracle_seed_distances(oracle_lib, q, s)
    assert np.array_equal(sd_g, sd_o)
    absolute = float(np.median(sd_o)) * 0.98  # roughly half of the queries fail the absolute test
    for kind, orc in ((capi.OUT_DMATCH, H.oracle_match_dmatch), (capi.OUT_UINT2_PAIR, H.oracle_match_pairs)):
        for rel in (0.6, 0.97, 1.1):
            g = run_gpu(capi, 0, q, t, kind, seed=sd_o, rel=rel, absolute=absolute)
            o = orc(oracle_lib, 0, 3, q, 7, t, None, None, 0, 0, sd_o, rel, absolute)
            if kind == capi.OUT_DMATCH:
                assert_dmatch_equal(g, o)
            else:
                assert np.array_equal(g["a"], o["a"]) and np.array_equal(g["b"], o["b"])
    # Match output kind (no distance field): the brute-force kernel compares with rel, not rel^2
    for seed in (None, sd_o):
        g = run_gpu(capi, 0, q, t, capi.OUT_MATCH, seed=seed, rel=0.97, absolute=absolute)
        o = H.oracle_match_match(oracle_lib, 0, 3, q, 7, t, None, None, 0, 0, seed, 0.97, absolute)
        assert np.array_equal(g["invalid"], o["invalid"])
        ok = o["invalid"] == 0
        assert np.array_equal(g["kp1_loc"][ok], o["kp1_loc"][ok]) and np.array_equal(g["kp0_loc"][ok], o["kp0_loc"][ok])


def test_empty_target_set_and_seed_flt_max(capi):
    q = random_features(10, 11)
    g = run_gpu(capi, 0, q, random_features(0, 12), capi.OUT_DMATCH)
    assert (g["invalid"] == 1).all() and (g["distance"] == np.float32(ABS)).all()
    sd = capi.seed_distances(capi.to_dev(q), len(q), None, 0).cpu().numpy()
    assert (sd == np.finfo(np.float32).max).all()


def test_double_constrained_matches_oracle(capi, oracle_lib):
    """Epipolar prefilter with the fixture cameras (GEO_ORBIT path of doFeatureMatching)."""
    v = H.load_view("Pipeline2View")
    cams = v["cameras"]
    proj = capi.projection_matrix(cams[1:2])
    q = random_features(3000, 13, hi=48)
    t = random_features(5000, 14, hi=48)
    for eps, delta in ((25.0, 5.0), (5.0, 0.0), (200.0, 50.0)):
        g = run_gpu(capi, 1, q, t, capi.OUT_DMATCH, eps=eps, delta=delta, cam=cams[0:1], proj=proj, absolute=3e7)
        o = H.oracle_match_dmatch(oracle_lib, 1, 3, q, 7, t, cams[0:1], proj, eps, delta, None, REL, 3e7)
        assert_dmatch_equal(g, o)
    assert 0 < (o["invalid"] == 0).sum()


def test_fundamental_constrained_matches_oracle(capi, oracle_lib):
    """matchFeaturesConstrained (src/MatchFactory.cu:1599-1657, :1710-1775, :1921-1980, :2127-2193, :2572-2626,
    :2762-2823): candidates within epsilon px of the epipolar line F (x, y, 1); all three output kinds with their
    ratio rules (Match and uint2_pair vs rel, DMatch vs rel^2)."""
    q = random_features(2000, 23, hi=48)
    t = random_features(3000, 24, hi=48)
    s = random_features(300, 25, hi=48)
    # a plausible F: rectified pair rotated by a few degrees (lines of moderate slope through the image)
    F = np.array([[0.0, -1e-4, 0.02], [1e-4, 0.0, -0.9], [-0.03, 1.0, 40.0]], np.float32)
    sd = H.oracle_seed_distances(oracle_lib, q, s)
    for eps in (3.0, 40.0, 800.0):
        for kind, orc in ((capi.OUT_DMATCH, H.oracle_match_dmatch), (capi.OUT_UINT2_PAIR, H.oracle_match_pairs),
                          (capi.OUT_MATCH, H.oracle_match_match)):
            for seed in (None, sd):
                params = capi.make_match_params(2, 3, 7, eps, 0.0, 0.9, 3e7, fundamental=F)
                out_d = capi.match(capi.to_dev(q), len(q), capi.to_dev(t), len(t), params, kind,
                                   seed_d=capi.to_dev(seed) if seed is not None else None)
                dt = {capi.OUT_DMATCH: H.DMATCH, capi.OUT_UINT2_PAIR: H.UINT2_PAIR, capi.OUT_MATCH: H.MATCH}[kind]
                g = capi.to_host(out_d, dt, len(q))
                o = orc(oracle_lib, 2, 3, q, 7, t, None, F.reshape(-1), eps, 0.0, seed, 0.9, 3e7)
                if kind == capi.OUT_DMATCH:
                    assert_dmatch_equal(g, o)
                elif kind == capi.OUT_UINT2_PAIR:
                    assert np.array_equal(g["a"], o["a"]) and np.array_equal(g["b"], o["b"])
                else:
                    assert np.array_equal(g["invalid"], o["invalid"])
                    ok = o["invalid"] == 0
                    assert np.array_equal(g["kp1_loc"][ok], o["kp1_loc"][ok])
        if eps == 3.0:
            assert 0 < (o["invalid"] == 0).sum() < len(q)  # the constraint bites


def test_band_culling_is_conservative_on_degenerate_geometry(capi, oracle_lib):
    """The spatially culled modes must give the oracle's answer whatever the locations: NaN / negative / huge
    coordinates, all features on one spot, clusters, fewer than one tile of queries or targets, epsilon 0 and an
    epsilon wider than the image."""
    v = H.load_view("Pipeline2View")
    cams = v["cameras"]
    proj = capi.projection_matrix(cams[1:2])
    F = np.array([[0.0, -1e-4, 0.02], [1e-4, 0.0, -0.9], [-0.03, 1.0, 40.0]], np.float32)
    rng = np.random.default_rng(77)

    def weird(n, seed):
        f = random_features(n, seed, hi=48)
        r = np.random.default_rng(seed)
        loc = r.uniform(0, 1024, (n, 2)).astype(np.float32)
        k = max(1, n // 8)
        loc[r.choice(n, k, replace=False)] = r.uniform(300, 310, (k, 2))        # a dense cluster
        loc[r.choice(n, max(1, n // 50), replace=False)] = np.nan
        loc[r.choice(n, max(1, n // 50), replace=False), 0] = -50.0
        loc[r.choice(n, max(1, n // 50), replace=False), 1] = 3.0e9
        f["loc"] = loc
        return f

    cases = [(weird(700, 1), weird(1500, 2)), (weird(5, 3), weird(2000, 4)), (weird(900, 5), weird(7, 6))]
    same = random_features(400, 9, hi=48)
    same["loc"] = np.float32(512.0)
    cases.append((same, same.copy()))
    for q, t in cases:
        for eps, delta in ((0.0, 0.0), (25.0, 5.0), (5000.0, 100.0)):
            g = run_gpu(capi, 1, q, t, capi.OUT_DMATCH, eps=eps, delta=delta, cam=cams[0:1], proj=proj, absolute=3e7)
            o = H.oracle_match_dmatch(oracle_lib, 1, 3, q, 7, t, cams[0:1], proj, eps, delta, None, REL, 3e7)
            assert_dmatch_equal(g, o)
            params = capi.make_match_params(2, 3, 7, eps, 0.0, REL, 3e7, fundamental=F)
            out_d = capi.match(capi.to_dev(q), len(q), capi.to_dev(t), len(t), params, capi.OUT_UINT2_PAIR)
            g2 = capi.to_host(out_d, H.UINT2_PAIR, len(q))
            o2 = H.oracle_match_pairs(oracle_lib, 2, 3, q, 7, t, None, F.reshape(-1), eps, 0.0, None, REL, 3e7)
            assert np.array_equal(g2["a"], o2["a"]) and np.array_equal(g2["b"], o2["b"])


def test_fundamental_constrained_steep_and_degenerate_lines(capi, oracle_lib):
    """F matrices whose epipolar lines are steep, nearly vertical, exactly vertical for some queries (b = 0: the
    reference's quotient is inf / NaN there) or fan out from an epipole inside the image: the bands are then far from
    the frame's direction, bounded by their hulls or not at all, and the result must still be the oracle's."""
    q = random_features(1500, 41, hi=48)
    t = random_features(2500, 42, hi=48)
    q["loc"][:40, 0] = 512.0  # with Fs[2] these queries get b = 0 exactly
    Fs = [np.array([[0.0, -1e-4, 0.9], [1e-4, 0.0, -0.05], [-400.0, 30.0, 100.0]], np.float32),    # steep lines
          np.array([[0.0, 0.0, 1.0], [0.0, 0.0, 1e-4], [-500.0, 0.0, 0.0]], np.float32),            # x = const (+ tiny slope)
          np.array([[0.0, 0.0, 1.0], [1.0, 0.0, -512.0], [-300.0, -1.0, 900.0]], np.float32),       # b = x - 512: zero for some
          np.array([[0.0, -1.0, 500.0], [1.0, 0.0, -500.0], [-500.0, 500.0, 0.0]], np.float32)]     # epipole at (500, 500)
    for F in Fs:
        for eps in (2.0, 30.0):
            params = capi.make_match_params(2, 3, 7, eps, 0.0, 0.9, 3e7, fundamental=F)
            out_d = capi.match(capi.to_dev(q), len(q), capi.to_dev(t), len(t), params, capi.OUT_UINT2_PAIR)
            g = capi.to_host(out_d, H.UINT2_PAIR, len(q))
            o = H.oracle_match_pairs(oracle_lib, 2, 3, q, 7, t, None, F.reshape(-1), eps, 0.0, None, 0.9, 3e7)
            assert np.array_equal(g["a"], o["a"]) and np.array_equal(g["b"], o["b"])


@pytest.mark.parametrize("direction,strip", [("37", "5"), ("90", "64"), ("-63.5", "1"), ("180", "16")])
def test_band_frame_is_only_an_order(direction, strip):
    """The frame of the band-culled modes (matcher.hip "band culling": direction u of the bands, strips across it) decides
    what is culled, never what is found.  A developer build lets the environment force u and the strip width; the
    oracle-parity tests of the culled modes are repeated in a child process with the frame turned away from the bands
    (37, 90, -63.5 degrees: bands steep or across the frame) and with strips of 1 to 64 pixels."""
    env = H.dev_env(SSRLCV_BAND_DIR=direction, SSRLCV_BAND_STRIP=strip)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "double_constrained or fundamental_constrained or conservative_on_degenerate"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_key_sort_orders_by_key_then_index(capi):
    """ssrlcv_hip_sort_keys_u32 (csrc/spatial_sort.hip: bucket by the key's high half, bitonic per bucket): the
    permutation is numpy's order by (bucket, key, index), which is the stable argsort of the keys while the strips fit one
    window of 4096 -- for frame-like keys (a few hundred strips), for arbitrary 32-bit words (every bucket used, strips
    aliasing into buckets), for one bucket holding everything (the global-memory network, sizes that are and are not
    powers of two), for duplicates and for tiny inputs."""
    import torch
    rng = np.random.default_rng(8)
    cases = []
    strips = rng.integers(32768 - 150, 32768 + 150, 300000).astype(np.uint32)
    cases.append((strips << 16) | rng.integers(32768 - 3000, 32768 + 3000, 300000).astype(np.uint32))   # frame-like
    cases.append(rng.integers(0, 2 ** 32, 200000, dtype=np.uint64).astype(np.uint32))                    # arbitrary words
    cases.append((np.uint32(32768) << 16) | rng.integers(0, 65536, 20000).astype(np.uint32))            # one bucket, > LDS
    cases.append((np.uint32(32768) << 16) | rng.integers(0, 65536, 16384).astype(np.uint32))            # ... a power of two
    cases.append((np.uint32(40000) << 16) | rng.integers(0, 4, 5000).astype(np.uint32))                 # heavy duplicates
    cases.append(np.full(777, 0xFFFFFFFF, np.uint32))
    cases.append(np.array([5], np.uint32))
    cases.append(np.array([9, 3, 3, 1 << 31, 0], np.uint32))
    for keys in cases:
        perm = capi.sort_keys(capi.to_dev(keys), len(keys)).cpu().numpy().view(np.uint32)
        bucket = ((keys >> 16) + 2048) & 4095
        ref = np.lexsort((np.arange(len(keys)), keys, bucket)).astype(np.uint32)
        assert np.array_equal(perm, ref), len(keys)
        if int(keys.max() >> 16) - int(keys.min() >> 16) < 2048 and 30720 <= int(keys.min() >> 16) and int(keys.max() >> 16) < 34816:
            assert np.array_equal(perm, np.argsort(keys, kind="stable").astype(np.uint32))
    assert capi.sort_keys(torch.empty(0, dtype=torch.uint8, device="cuda"), 0).numel() == 0


def test_compact_matches_is_stable(capi):
    q, t = random_features(5000, 15, hi=64), random_features(800, 16, hi=64)
    probe = run_gpu(capi, 0, q, t, capi.OUT_DMATCH, absolute=3e7)
    absolute = float(np.median(probe["distance"]))  # about half of the matches become invalid
    g_d = capi.match(capi.to_dev(q), len(q), capi.to_dev(t), len(t),
                     capi.make_match_params(0, 0, 1, 0, 0, REL, absolute), capi.OUT_DMATCH)
    before = capi.to_host(g_d, H.DMATCH, len(q))
    ws = capi.match_workspace(len(q), len(t))
    n = capi.compact_matches(capi.OUT_DMATCH, g_d, len(q), ws)
    after = capi.to_host(g_d, H.DMATCH, n)
    keep = before[before["invalid"] == 0]
    assert 0 < n == len(keep) < len(q)
    assert np.array_equal(after["kp0_loc"], keep["kp0_loc"]) and np.array_equal(after["distance"], keep["distance"])


def test_everest_fixture_matches_reproduced_on_gpu(capi, oracle_lib, everest_oracle_features):
    """The reference's FeatureMatching2View golden output, matched on the GPU from the oracle's SIFT features."""
    f0, f1, _ = everest_oracle_features
    seed, _ = H.load_seed_features()
    v = H.load_view("Pipeline2View")
    cams = v["cameras"]
    f0_d, f1_d = capi.to_dev(f0), capi.to_dev(f1)
    sd_d = capi.seed_distances(f0_d, len(f0), capi.to_dev(seed), len(seed))
    assert np.array_equal(sd_d.cpu().numpy(), H.oracle_seed_distances(oracle_lib, f0, seed))
    params = capi.make_match_params(1, 0, 1, 25.0, 5.0, REL, ABS, cams[0:1], capi.projection_matrix(cams[1:2]))
    out_d = capi.match(f0_d, len(f0), f1_d, len(f1), params, capi.OUT_DMATCH, seed_d=sd_d)
    ws = capi.match_workspace(len(f0), len(f1))
    n = capi.compact_matches(capi.OUT_DMATCH, out_d, len(f0), ws)
    dm = capi.to_host(out_d, H.DMATCH, n)
    kp = v["kp0"]
    assert n == 13534
    assert np.array_equal(dm["kp0_loc"], kp["loc"][0::2]) and np.array_equal(dm["kp1_loc"], kp["loc"][1::2])
    assert np.array_equal(dm["kp0_parent"], kp["parentId"][0::2])
    # M7 on the device (src/Pipeline.cu:198-224): the validated list laid out pairwise is the reference's golden MatchSet
    kp_d, mm_d, peak = capi.matchset_from_matches(capi.OUT_DMATCH, out_d, n, want_max=True)
    gkp, gmm = capi.to_host(kp_d, H.KEYPOINT, 2 * n), capi.to_host(mm_d, H.MULTIMATCH, n)
    assert np.array_equal(gkp["loc"], kp["loc"]) and np.array_equal(gkp["parentId"], kp["parentId"])
    assert np.array_equal(gmm["numKeyPoints"], v["mm0"]["numKeyPoints"]) and np.array_equal(gmm["index"], v["mm0"]["index"])
    assert peak == float(dm["distance"].max())


@pytest.mark.parametrize("n", [0, 1, 257, 5000])
def test_matchset_from_matches_both_input_kinds(capi, n):
    rng = np.random.default_rng(n)
    for kind, dt in ((capi.OUT_DMATCH, H.DMATCH), (capi.OUT_MATCH, H.MATCH)):
        m = np.zeros(n, dt)
        m["kp0_parent"], m["kp1_parent"] = 2, 5
        m["kp0_loc"] = rng.uniform(0, 4096, (n, 2)).astype(np.float32)
        m["kp1_loc"] = rng.uniform(0, 4096, (n, 2)).astype(np.float32)
        if kind == capi.OUT_DMATCH:
            m["distance"] = rng.integers(0, 40000, n).astype(np.float32)
        want_max = kind == capi.OUT_DMATCH
        kp_d, mm_d, peak = capi.matchset_from_matches(kind, capi.to_dev(m) if n else None, n, want_max=want_max)
        kp, mm = capi.to_host(kp_d, H.KEYPOINT, 2 * n), capi.to_host(mm_d, H.MULTIMATCH, n)
        assert np.array_equal(kp["loc"][0::2], m["kp0_loc"]) and np.array_equal(kp["loc"][1::2], m["kp1_loc"])
        assert np.array_equal(kp["parentId"][0::2], m["kp0_parent"]) and np.array_equal(kp["parentId"][1::2], m["kp1_parent"])
        assert np.all(mm["numKeyPoints"] == 2) and np.array_equal(mm["index"], 2 * np.arange(n))
        if want_max:
            assert peak == (float(m["distance"].max()) if n else 0.0)
