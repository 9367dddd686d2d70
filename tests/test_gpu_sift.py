"""GPU parity: SIFT leg (pyramid, key points, orientations, descriptors) through the C ABI vs the CPU oracle.

Everything is bit-exact: the pure-arithmetic stages (S1-S12) always were; the stages that call elementary functions
(powf in refinement's sigma, atan2f / expf in orientation, sinf / cosf / expf / atan2f in descriptors) use the functions
of ssrlcv_amd/csrc/sv_math.h, the same source text as the oracle's oracle/oracle_libm.h, and the kernels replay the
reference's operation order (sequential orientation histogram, vote association, IEEE divisions).
"""
import ctypes

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from ssrlcv_amd import capi
    return capi


@pytest.fixture(scope="module")
def image_small():
    return H.synthetic_image(384, 256, seed=1)


def test_image_ops_bit_exact(capi, oracle_lib):
    rng = np.random.default_rng(0)
    w, h = 200, 96
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    f = capi.u8_to_f32(capi.to_dev(img), w * h).cpu().numpy()
    assert np.array_equal(f, img.reshape(-1).astype(np.float32))
    up_o = np.zeros((2 * h, 2 * w), np.float32)
    oracle_lib.oracle_upsample2x(H.P(img.astype(np.float32)), ctypes.c_uint32(w), ctypes.c_uint32(h), H.P(up_o))
    up_g = capi.upsample2x_u8(capi.to_dev(img), w, h).cpu().numpy().reshape(2 * h, 2 * w)
    assert np.array_equal(up_g, up_o)
    up_g2 = capi.upsample2x(capi.to_dev(img.astype(np.float32)), w, h).cpu().numpy().reshape(2 * h, 2 * w)
    assert np.array_equal(up_g2, up_o)
    src = (rng.standard_normal((h, w)) * 50 + 100).astype(np.float32)
    bin_o = np.zeros((h // 2, w // 2), np.float32)
    oracle_lib.oracle_bin2x(H.P(src), ctypes.c_uint32(w), ctypes.c_uint32(h), H.P(bin_o))
    assert np.array_equal(capi.bin2x(capi.to_dev(src), w, h).cpu().numpy().reshape(h // 2, w // 2), bin_o)
    mm = capi.minmax(capi.to_dev(src), w * h).cpu().numpy()
    assert mm[0] == src.min() and mm[1] == src.max()


def test_normalize_bit_exact(capi):
    """normalize (src/Image.cu:1560-1565) = (x - min) / (max - min) with IEEE division: the kernels divide through
    sv::div_by (shared refined reciprocal + the fma chain of the compiler's own expansion), which must return the
    correctly rounded quotient numpy's float32 division returns -- on random data, on the end points, on values one
    ulp from them and on level-like ranges (tiny and large spans)."""
    import torch
    rng = np.random.default_rng(7)
    for lo, hi in [(-0.37, 0.81), (0.0, 255.0), (-3.1e-3, 2.7e-3), (1.0e-3, 1.0001e-3), (-7.5e4, 9.1e5)]:
        x = rng.uniform(lo, hi, 1 << 20).astype(np.float32)
        mn, mx = np.float32(x.min()), np.float32(x.max())
        x[:4] = [mn, mx, np.nextafter(mn, mx, dtype=np.float32), np.nextafter(mx, mn, dtype=np.float32)]
        ref = (x - mn) / (mx - mn)
        d = capi.to_dev(x)
        capi.normalize_(d, x.size, capi.to_dev(np.array([mn, mx], np.float32)))
        got = d.cpu().numpy().view(np.float32).reshape(-1)[: x.size]
        assert np.array_equal(got, ref), (lo, hi, int((got != ref).sum()))


@pytest.mark.parametrize("w,h", [(512, 264), (64, 64), (1000, 72), (2048, 136), (250, 130), (1002, 96)])
@pytest.mark.parametrize("sigma,pw", [(0.70710678, 0.5), (1.0, 0.5), (1.4142135, 0.5), (2.0, 0.5), (2.828427, 0.5),
                                      (4.0, 0.5), (1.3, 1.0)])
def test_separable_gaussian_bit_exact(capi, oracle_lib, w, h, sigma, pw):
    """Every templated radius (taps 13,17,23,33,47,65) + a padded odd one (taps 11), strips narrower/wider than 256
    columns, heights that are not multiples of the 8-row marching step, mirrored borders closer than the radius."""
    taps, wgt = capi.gauss_kernel(sigma, pw)
    otaps, owgt = H.oracle_gauss_kernel(oracle_lib, sigma, pw)
    assert taps == otaps and np.array_equal(wgt, owgt)
    if taps // 2 >= h:
        pytest.skip("radius exceeds the mirrored range of this test image")
    rng = np.random.default_rng(w * 7 + h)
    src = (rng.standard_normal((h, w)) * 40 + 120).astype(np.float32)
    ref = np.zeros((h, w), np.float32)
    oracle_lib.oracle_conv_separable(H.P(src), ctypes.c_uint32(w), ctypes.c_uint32(h), ctypes.c_int(taps), H.P(wgt),
                                     H.P(ref))
    out_d, mm_d = capi.gauss_sep_conv(capi.to_dev(src), w, h, wgt)
    out = out_d.cpu().numpy().reshape(h, w)
    assert np.array_equal(out, ref)
    mm = mm_d.cpu().numpy()
    assert mm[0] == ref.min() and mm[1] == ref.max()


@pytest.mark.parametrize("size", [(384, 256), (256, 256), (262, 260), (300, 258), (257, 263), (261, 264)])
def test_dog_pyramid_bit_exact(capi, oracle_lib, size):
    """Sizes 3-6 go through makeBinnable (S3): even sizes padded to multiples of 8 before the upsample (one or both
    sides), sizes with an odd side padded to multiples of 32 after it."""
    w, h = size
    img = H.synthetic_image(w, h, seed=2)
    osf = H.OracleSift(oracle_lib, img)
    plan = capi.SiftPlan(w, h)
    plan.build_dog(capi.to_dev(img))
    try:
        for o in range(4):
            ow, oh, pw, sig = osf.octave_info(o)
            for b in range(5):
                lvl, (mn, mx) = plan.level(0, o, b)
                assert lvl.shape == (oh, ow)
                ref = osf.level(1, o, b)
                assert np.array_equal(lvl, ref), (o, b, np.abs(lvl - ref).max())
                omn, omx = osf.minmax(1, o, b)
                assert (mn, mx) == (omn, omx)
        # gaussian levels of the last octave are still in the workspace (un-normalised): compare after normalising
        for b in range(6):
            lvl, (mn, mx) = plan.level(1, 3, b)
            assert (mn, mx) == osf.minmax(0, 3, b)
            assert np.array_equal((lvl - np.float32(mn)) / (np.float32(mx) - np.float32(mn)), osf.level(0, 3, b))
    finally:
        osf.close()


def _compare_keypoints(g, o, stage):
    assert len(g) == len(o), (stage, len(g), len(o))
    for name in ("octave", "blur"):
        assert np.array_equal(g[name], o[name]), (stage, name)
    if stage < 2:
        assert np.array_equal(g["loc"], o["loc"]) and np.array_equal(g["intensity"], o["intensity"])
        assert np.array_equal(g["sigma"], o["sigma"])
    else:
        # refinement is +-*/ and the shared powf (sv_math.h): everything bit-equal
        assert np.array_equal(g["loc"], o["loc"]), stage
        assert np.array_equal(g["intensity"], o["intensity"]), stage
        assert np.array_equal(H.bits(g["sigma"]), H.bits(o["sigma"])), stage
    if stage >= 6:  # the reference's sequential histogram chain + shared expf / atan2f: thetas bit-equal
        assert np.array_equal(H.bits(g["theta"]), H.bits(o["theta"])), stage


@pytest.mark.parametrize("stage", [0, 1, 2, 3, 4, 5, 6])
def test_keypoint_stages_match_oracle(capi, oracle_lib, image_small, stage):
    """searchForExtrema -> removeNoise -> refine (+sort, re-scan) -> removeNoise -> removeEdges -> checkKeyPoints ->
    orientations: list contents, order and extremaBlurIndices after every stage."""
    img = image_small
    h, w = img.shape
    osf = H.OracleSift(oracle_lib, img)
    plan = capi.SiftPlan(w, h)
    plan.build_dog(capi.to_dev(img))
    plan.set_stop_stage(stage)
    plan.describe()
    okps, oidx = osf.keypoints(stage)
    osf.close()
    pos = 0
    total = 0
    for o in range(4):
        g, gidx, overflow = plan.keypoints(o, H.SSKEYPOINT)
        assert overflow == 0
        n_o = int(oidx[o][5])
        _compare_keypoints(g, okps[pos: pos + n_o], stage)
        if n_o:
            assert np.array_equal(gidx[:5], oidx[o][:5]), (o, gidx, oidx[o])
        pos += n_o
        total += len(g)
    assert total == len(okps) and total > 50
    assert plan.count() == total


@pytest.mark.parametrize("size", [None, (262, 260), (257, 263), (256, 129), (264, 137), (776, 200)])
def test_features_match_oracle(capi, oracle_lib, image_small, size):
    """None = the 384 x 256 image; the next two sizes run with makeBinnable's zero border (even / odd branch):
    locations are in the padded frame on both sides; the last three leave the fused DoG pass's last row segment with one, two
    and a few rows (16-row segments over 2 x 129 / 2 x 137 rows; 22-row ones further down), i.e. short last trips of its
    three-row loop and of the LDS-DMA row ring."""
    img = image_small if size is None else H.synthetic_image(size[0], size[1], seed=5)
    h, w = img.shape
    of = H.oracle_sift(oracle_lib, img)
    plan = capi.SiftPlan(w, h)
    plan.extract(capi.to_dev(img))
    gf = plan.features_host(H.FEATURE)
    assert len(gf) == len(of) > 100
    assert (gf["parent"] == -1).all()
    H.assert_features_equal(gf, of)  # loc, sigma, theta and all 128 descriptor bytes, bit for bit


def test_everest_end_to_end_reproduces_reference_matches(capi, oracle_lib):
    """HIP SIFT on the reference's 1024x1024 pixel fixtures + HIP seed distances + HIP constrained matcher +
    compaction + HIP triangulation -> the reference's golden key points (bit-exact) and cloud (RMS <= 1e-4 km)."""
    pix = H.load_everest_pixels()
    seed, _ = H.load_seed_features()
    v = H.load_view("Pipeline2View")
    cams = v["cameras"]
    plans = [capi.SiftPlan(1024, 1024), capi.SiftPlan(1024, 1024)]
    for i in range(2):
        plans[i].extract(capi.to_dev(pix[i]))
    n0, n1 = plans[0].count(), plans[1].count()
    assert (n0, n1) == (32425, 36251)  # the oracle's counts on these images
    f0_d, f1_d = plans[0].features, plans[1].features
    sd_d = capi.seed_distances(f0_d, n0, capi.to_dev(seed), len(seed))
    params = capi.make_match_params(1, 0, 1, 25.0, 5.0, 0.6, 200.0 * 200.0, cams[0:1],
                                    capi.projection_matrix(cams[1:2]))
    out_d = capi.match(f0_d, n0, f1_d, n1, params, capi.OUT_DMATCH, seed_d=sd_d)
    n = capi.compact_matches(capi.OUT_DMATCH, out_d, n0, capi.match_workspace(n0, n1))
    dm = capi.to_host(out_d, H.DMATCH, n)
    kp = v["kp0"]
    ref_pairs = {(tuple(a), tuple(b)) for a, b in zip(kp["loc"][0::2].tolist(), kp["loc"][1::2].tolist())}
    got_pairs = {(tuple(a), tuple(b)) for a, b in zip(dm["kp0_loc"].tolist(), dm["kp1_loc"].tolist())}
    # the reference's golden match list, entry for entry: 13 534 matches, same order, every location bit-equal
    print("2-view golden: HIP %d matches, %d / %d golden pairs" % (n, len(ref_pairs & got_pairs), len(ref_pairs)))
    assert n == 13534 and got_pairs == ref_pairs
    assert np.array_equal(dm["kp0_loc"], kp["loc"][0::2]) and np.array_equal(dm["kp1_loc"], kp["loc"][1::2])
    assert np.array_equal(dm["kp0_parent"], kp["parentId"][0::2]) and np.array_equal(dm["kp1_parent"], kp["parentId"][1::2])
    # triangulate the reference's own match set on the GPU -> golden cloud
    b_d, l_d = capi.generate_bundles(capi.to_dev(v["mm0"]), capi.to_dev(kp), len(v["mm0"]), capi.to_dev(cams), 2, len(kp))
    pts_d, _, _ = capi.triangulate(l_d, b_d, len(v["mm0"]))
    diff = pts_d.cpu().numpy().reshape(-1, 3) - v["points0"]
    assert float(np.sqrt((diff.astype(np.float64) ** 2).sum(1).mean())) == 0.0


def test_nview_flow_single_rank_matches_3view_fixture(capi):
    """ssrlcv_amd.pipeline.reconstruct (the multi-GPU driver at world size 1) on the three everest fixtures:
    MultiMatch structure and cloud vs Pipeline3View golden outputs."""
    from ssrlcv_amd import pipeline
    pix = [capi.to_dev(p).view(1024, 1024) for p in H.load_everest_pixels()]
    seed, _ = H.load_seed_features()
    v = H.load_view("Pipeline3View")
    res = pipeline.reconstruct(pix, v["cameras"], seed_features=seed)
    mm, kp = res["matches"], res["keypoints"]
    ref_mm, ref_kp = v["mm0"], v["kp0"]
    # the reference's golden MatchSet, entry for entry: 21 177 multi-matches, 51 442 key points
    assert len(mm) == len(ref_mm) == 21177
    assert np.array_equal(mm["numKeyPoints"], ref_mm["numKeyPoints"]) and np.array_equal(mm["index"], ref_mm["index"])
    assert np.array_equal(kp["parentId"], ref_kp["parentId"]) and np.array_equal(kp["loc"], ref_kp["loc"])
    pts = res["points"].cpu().numpy()
    assert pts.shape == (len(mm), 3) and np.isfinite(pts).all()
    diff = pts - v["points0"]
    assert float(np.sqrt((diff.astype(np.float64) ** 2).sum(1).mean())) == 0.0  # every point bit-equal to the reference's cloud (round 4)


@pytest.mark.parametrize("kind", ["const", "zeros", "rand", "checker", "onepixel"])
def test_degenerate_images_match_oracle(capi, oracle_lib, kind):
    """Edge inputs: a constant image (every normalisation is 0/0: no extrema, no features, no fault), an all-zero image
    of a size makeBinnable pads, uniform noise on a padded size (feature-dense), a one-pixel checkerboard and a single
    bright pixel.  Every feature field is the oracle's, bit for bit."""
    img = {
        "const": np.full((256, 256), 128, np.uint8),
        "zeros": np.zeros((264, 300), np.uint8),
        "rand": np.random.default_rng(1).integers(0, 256, (258, 262), dtype=np.uint8),
        "checker": ((np.indices((256, 384)).sum(0) & 1) * 255).astype(np.uint8),
        "onepixel": np.pad(np.full((1, 1), 255, np.uint8), ((128, 127), (200, 183))),
    }[kind]
    h, w = img.shape
    plan = capi.SiftPlan(w, h)
    plan.extract(capi.to_dev(img))
    gf = plan.features_host(H.FEATURE)
    of = H.oracle_sift(oracle_lib, img)
    assert len(gf) == len(of), (kind, len(gf), len(of))
    # bit for bit, the single pixel included: its mirror symmetry leaves exactly tied histogram peaks, and which of two
    # tied bins wins is decided by the last bit of expf / atan2f -- the same bit on both sides now (sv_math.h)
    H.assert_features_equal(gf, of)


_CONV_SCRIPT = r"""
import ctypes, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
import helpers as H
from ssrlcv_amd import capi
lib = H.oracle()
n = 0
for (w, h) in [(512, 264), (1024, 200), (256, 1100), (384, 136), (1000, 72)]:
    for sigma in (0.70710678, 1.0, 1.4142135, 2.0, 2.828427, 4.0):
        taps, wgt = capi.gauss_kernel(sigma, 0.5)
        rng = np.random.default_rng(w + h + taps)
        src = (rng.standard_normal((h, w)) * 40 + 120).astype(np.float32)
        ref = np.zeros((h, w), np.float32)
        lib.oracle_conv_separable(H.P(src), ctypes.c_uint32(w), ctypes.c_uint32(h), ctypes.c_int(taps), H.P(wgt), H.P(ref))
        out_d, mm_d = capi.gauss_sep_conv(capi.to_dev(src), w, h, wgt)
        out = out_d.cpu().numpy().reshape(h, w)
        assert np.array_equal(out, ref), (w, h, taps, int((out != ref).sum()))
        mm = mm_d.cpu().numpy()
        assert mm[0] == ref.min() and mm[1] == ref.max(), (w, h, taps)
        n += 1
print("CONV OK", n)
"""


@pytest.mark.parametrize("variant", [
    {"SSRLCV_GAUSS_TILE_MAXPX": "0"},                                   # marching kernels: strips, k_gauss_mfma2 (128 columns)
    {"SSRLCV_GAUSS_TILE_MAXPX": "0", "SSRLCV_GAUSS_WIDE": "1"},         # 256-column strips
    {"SSRLCV_GAUSS_TILE_MAXPX": "0", "SSRLCV_GAUSS_MFMA_MINR": "6"},    # MFMA for every radius
    {"SSRLCV_GAUSS_TILE_MAXPX": "0", "SSRLCV_GAUSS_MFMA_MINR": "6", "SSRLCV_GAUSS_WIDE": "1"},
    {"SSRLCV_GAUSS_TILE_MAXPX": "0", "SSRLCV_GAUSS_VALU": "1"},         # VALU formulation for every radius
    {"SSRLCV_GAUSS_TILE_MAXPX": "0", "SSRLCV_GAUSS_MFMA_MINR": "6", "SSRLCV_GAUSS_NARROW": "1"},  # 128-column MFMA strips + a VALU remainder strip
    {"SSRLCV_GAUSS_TILE_MAXPX": "0", "SSRLCV_GAUSS_RM": "63", "SSRLCV_GAUSS_RM_MINPX": "0"},      # register-marching 4x4x1 kernel for every radius
    {"SSRLCV_GAUSS_TILE_MAXPX": "0", "SSRLCV_GAUSS_RM": "63", "SSRLCV_GAUSS_RM_MINPX": "0", "SSRLCV_GAUSS_RM_ROWS": "48"},  # ... in short blocks
], ids=lambda v: "+".join(k.replace("SSRLCV_GAUSS_", "") + "=" + x for k, x in v.items()))
def test_every_gaussian_formulation_is_bit_exact(variant):
    """The formulation is chosen per process (environment) and by level size: the default suite reaches the tile kernel
    on its small images; the marching kernels (VALU strips, both MFMA editions, 128- and 256-column strips) are run here
    and the register-marching kernel in child processes on sizes with full and partial strips, each against the oracle bit for
    bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = H.dev_env(**variant)
    r = subprocess.run([sys.executable, "-c", _CONV_SCRIPT % {"root": root}], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "CONV OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


_PYRAMID_SCRIPT = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
import helpers as H
from ssrlcv_amd import capi
lib = H.oracle()
for (w, h) in [(512, 384), (262, 260)]:
    img = H.synthetic_image(w, h, seed=5)
    osf = H.OracleSift(lib, img)
    plan = capi.SiftPlan(w, h)
    plan.build_dog(capi.to_dev(img))
    for o in range(4):
        for b in range(5):
            lvl, (mn, mx) = plan.level(0, o, b)
            ref = osf.level(1, o, b)
            assert np.array_equal(lvl, ref), (w, h, o, b, int((lvl != ref).sum()))
            assert (mn, mx) == osf.minmax(1, o, b), (w, h, o, b)
    # the extrema the fused DoG pass flagged (raw, and past the first removeNoise): stages 0 and 1 of the key-point lists
    for stage in (0, 1):
        plan.set_stop_stage(stage)
        plan.describe()
        okps, oidx = osf.keypoints(stage)
        pos = 0
        for o in range(4):
            g, gidx, overflow = plan.keypoints(o, H.SSKEYPOINT)
            n_o = int(oidx[o][5])
            ref = okps[pos: pos + n_o]
            assert overflow == 0 and len(g) == n_o, (w, h, stage, o, len(g), n_o)
            for name in ("octave", "blur", "loc", "intensity", "sigma"):
                assert np.array_equal(g[name], ref[name]), (w, h, stage, o, name)
            pos += n_o
    osf.close()
print("PYRAMID OK")
"""


@pytest.mark.parametrize("variant", [
    {"SSRLCV_NO_BIN_FUSION": "1"},                                  # 2x2 bin by k_bin2x instead of the level-3 convolution
    {"SSRLCV_DOG_SPLIT": "1"},                                      # the DoG / extrema pass split on every octave (levels 0-3 early, 1-5 late)
    {"SSRLCV_DOGX_WAVES": "65536"},                                 # the shortest row segments
    {"SSRLCV_DOGX_WAVES": "2048"},                                  # long row segments (many trips of the LDS-DMA row ring)
    {"SSRLCV_DOGX0_AFTER": "1"},                                    # octave 0's DoG pass held back behind octave 1's level 3
    {"SSRLCV_DOGX_NPX": "1", "SSRLCV_DOGX_WAVES": "512"},           # one pixel per lane, long row segments
    {"SSRLCV_DOGX_NPX": "2", "SSRLCV_DOG_SPLIT": "1"},              # two pixels per lane
    {"SSRLCV_GAUSS_TILE_MAXPX": "0", "SSRLCV_SIFT_SERIAL": "1"},    # marching kernels only, one stream
    {"SSRLCV_GAUSS_TILE_MAXPX": "100000000"},                       # tile kernel for every level
    {"SSRLCV_GAUSS_TILE_MAXPX": "0", "SSRLCV_GAUSS_RM": "63", "SSRLCV_GAUSS_RM_MINPX": "0"},  # register-marching Gaussian (with the folded 2x2 bin)
    {"SSRLCV_GAUSS_TILE_MAXPX": "0", "SSRLCV_GAUSS_RM": "63", "SSRLCV_GAUSS_RM_MINPX": "0", "SSRLCV_GAUSS_RM_ONEB": "63"},  # ... its one-barrier form (round 6)
    {"SSRLCV_GAUSS_PAIR_MINPX": "0"},                               # round 6: levels 0 + 1 of octave 0 in one launch on the matrix pipe (1024 x 768: five strips, the last one partial; top, bottom, left and right mirrors)
    {"SSRLCV_GAUSS_PAIR_MINPX": "0", "SSRLCV_GAUSS_PAIR_FORM": "valu"},  # ... and its vector formulation
    {"SSRLCV_NO_GAUSS_PAIR": "1"},                                  # the two launches
    {"SSRLCV_NO_XCD_STRIPS": "1"},                                  # strips in plain block order (the default gives an XCD neighbouring strips)
    {"SSRLCV_NO_XCD_STRIPS": "1", "SSRLCV_GAUSS_TILE_MAXPX": "0", "SSRLCV_GAUSS_RM": "63", "SSRLCV_GAUSS_RM_MINPX": "0"},
], ids=lambda v: "+".join(k.replace("SSRLCV_", "") + "=" + x for k, x in v.items()))
def test_every_pyramid_schedule_is_bit_exact(variant):
    """build_dog's developer switches (who makes the 2x2 bin, how the fused DoG / extrema pass is cut into launches, row
    segments and pixels per lane, which Gaussian kernel a level takes) change the schedule, never a bit: every DoG level
    (materialised on request from the Gaussian levels the workspace keeps), the {min, max} the fused pass reduced and the
    extrema it flagged against the oracle, in child processes (the switches are read once per process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = H.dev_env(**variant)
    r = subprocess.run([sys.executable, "-c", _PYRAMID_SCRIPT % {"root": root}], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "PYRAMID OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


_FEATURES_SCRIPT = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
import helpers as H
from ssrlcv_amd import capi
lib = H.oracle()
for (w, h, seed) in [(512, 384, 5), (262, 260, 6), (640, 128, 7)]:
    img = H.synthetic_image(w, h, seed=seed)
    plan = capi.SiftPlan(w, h)
    for rep in range(2):                       # twice on one plan: the group control words are re-armed per call
        plan.extract(capi.to_dev(img))
        g = plan.features_host(H.FEATURE)
    o = H.oracle_sift(lib, img)
    assert len(g) == len(o) and len(g) > 100, (w, h, len(g), len(o))
    H.assert_features_equal(g, o)
print("FEATURES OK")
"""


@pytest.mark.parametrize("variant", [
    {"SSRLCV_SAMPLING_PIPELINED": "1"},                                    # orientation / descriptor launches pipelined over four sampling groups
    {"SSRLCV_SAMPLING_PIPELINED": "1", "SSRLCV_SAMPLING_IRREGULAR": "1"},  # ... its fallback for blur indices that are not an ordered partition
    {"SSRLCV_EARLY_POLAR": "1"},                                           # gradient tables started from inside build_dog (fused extract)
    {"SSRLCV_EARLY_POLAR": "1", "SSRLCV_SAMPLING_PIPELINED": "1"},
    {"SSRLCV_NO_EARLY_CHAIN": "1"},                                        # octave 0's list chain in describe instead of behind its DoG pass
    {"SSRLCV_PHASED": "1"}, {"SSRLCV_PHASED": "2"},                        # build_dog: the octave chain on one stream, levels 4-5 + DoG passes beside it
    {"SSRLCV_THETAS_LANES": "1"}, {"SSRLCV_THETAS_LANES": "2"}, {"SSRLCV_THETAS_LANES": "4"},  # round 6: lanes that share a key point's orientation window
    {"SSRLCV_THETAS_LANES": "2", "SSRLCV_SAMPLING_PIPELINED": "1"},
    {"SSRLCV_THETAS_SPLIT": "1"},                                          # ... the small octaves' tables first, their orientations beside octave 0's tables
], ids=lambda v: "+".join(k.replace("SSRLCV_", "") + "=" + x for k, x in v.items()))
def test_every_sampling_schedule_is_bit_exact(variant):
    """Round 5's schedule experiments (developer build; exact, measured, not the defaults: profiles/r05_schedule_ab.txt): the
    orientation / descriptor kernels pipelined over four sampling groups (keypoints.hip: octave 0's expansion in three
    pieces that continue each other) with the fallback the device takes when an octave's blur indices are irregular, the
    gradient tables started from inside build_dog, and the phased order of build_dog's launches give the default schedule's
    features -- every field equal to the oracle's -- in child processes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _FEATURES_SCRIPT % {"root": root}], env=H.dev_env(**variant), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "FEATURES OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("size", [(64, 64), (72, 200), (136, 64), (200, 4096)])
def test_small_and_thin_images_match_oracle(capi, oracle_lib, size):
    """The reference takes any image of at least 8 x 8 pixels (src/FeatureFactory.cu:341-345,364-376); below 64 pixels its
    65-tap mirror reads outside the smallest octave, so 64 is where defined behaviour starts.  Levels with a side under 64
    pixels go through the one-tile Gaussian kernel with the reference's modulo mirror."""
    w, h = size
    img = H.synthetic_image(w, h, seed=9)
    of = H.oracle_sift(oracle_lib, img)
    plan = capi.SiftPlan(w, h)
    plan.extract(capi.to_dev(img))
    gf = plan.features_host(H.FEATURE)
    print("%d x %d: %d features" % (w, h, len(gf)))
    assert len(gf) == len(of)
    H.assert_features_equal(gf, of)


def test_stage_at_a_time_entry_point_matches_the_fused_call(capi, oracle_lib, image_small):
    """ssrlcv_hip_sift_stage runs one reference launch site per call (INTEGRATION.md option B).  After every call the
    octave lists equal the oracle's at that stage, and stages 0..7 in order land on the features of the fused
    ssrlcv_hip_sift_extract, bit for bit."""
    img = image_small
    h, w = img.shape
    osf = H.OracleSift(oracle_lib, img)
    plan = capi.SiftPlan(w, h)
    plan.build_dog(capi.to_dev(img))
    for stage in range(8):
        plan.stage(stage)
        if stage <= 6:
            okps, oidx = osf.keypoints(stage)
            pos = 0
            for o in range(4):
                g, gidx, overflow = plan.keypoints(o, H.SSKEYPOINT)
                n_o = int(oidx[o][5])
                assert overflow == 0
                _compare_keypoints(g, okps[pos: pos + n_o], stage)
                if n_o:
                    assert np.array_equal(gidx[:5], oidx[o][:5]), (stage, o, gidx, oidx[o])
                pos += n_o
            assert plan.count() == len(okps)
    osf.close()
    staged = plan.features_host(H.FEATURE)
    fused = capi.SiftPlan(w, h)
    fused.extract(capi.to_dev(img))
    H.assert_features_equal(staged, fused.features_host(H.FEATURE))
    H.assert_features_equal(staged, H.oracle_sift(oracle_lib, img))

