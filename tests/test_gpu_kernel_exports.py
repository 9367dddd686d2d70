"""The key-point stage one kernel at a time over caller-owned buffers (include/ssrlcv_hip.h): every
export stands for one launch site of src/FeatureFactory.cu / src/SIFT_FeatureFactory.cu and is held here to the oracle's
list after the corresponding stage (oracle_sift_keypoints), driven the way upstream's host code drives the kernel it
replaces: DoG images as Octave::blurs[b]->pixels holds them at that point, extremaBlurIndices on the host."""
import ctypes

import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu

u32, f32, ci = ctypes.c_uint32, ctypes.c_float, ctypes.c_int
NOISE, EDGE = 0.01, 12.1


@pytest.fixture(scope="module")
def ctx(oracle_lib):
    from ssrlcv_amd import capi
    img = H.synthetic_image(384, 320, seed=23)
    s = H.OracleSift(oracle_lib, img)
    stages = {k: s.keypoints(k) for k in range(7)}
    yield capi, s, stages
    s.close()


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()


def _ws(capi, n):
    capi.LIB.ssrlcv_hip_compact_workspace_bytes.restype = ctypes.c_size_t
    return capi.dev_bytes(int(capi.LIB.ssrlcv_hip_compact_workspace_bytes(u32(max(n, 1)))))


def _compact(capi, fn, t, n):
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    ws = _ws(capi, n)
    capi.check(fn(capi.ptr(t), u32(n), capi.ptr(cnt), capi.ptr(ws), ctypes.c_size_t(ws.numel()), capi.stream_ptr()))
    return int(cnt.item())


def _octave_slice(stage_list, o):
    kps, idx = stage_list
    start = int(sum(idx[k][5] for k in range(o)))
    return kps[start:start + int(idx[o][5])], idx[o]


def _same(a, b, fields=("octave", "blur", "loc", "intensity", "sigma", "theta")):
    assert len(a) == len(b), (len(a), len(b))
    for f in fields:
        assert np.array_equal(a[f].view(np.uint32) if a[f].dtype == np.float32 else a[f],
                              b[f].view(np.uint32) if b[f].dtype == np.float32 else b[f]), f


def _levels(s, kind, o):
    return [s.level(kind, o, b) for b in range(5)]


@pytest.mark.parametrize("o", [0, 1, 2, 3])
def test_search_for_extrema_exports(ctx, o):
    """findExtrema + thrust::remove + fillExtrema per blur (src/FeatureFactory.cu:86-159) == the oracle's stage-0 list."""
    capi, s, stages = ctx
    w, h, pw, sig = s.octave_info(o)
    raw = [_dev(l) for l in _levels(s, 1, o)]          # the DoG levels before findKeyPoints normalises them again
    out = []
    for b in (1, 2, 3):
        addr = torch.full((w * h,), -1, dtype=torch.int32, device="cuda")
        capi.check(capi.LIB.ssrlcv_hip_find_extrema(u32(w), u32(h), capi.ptr(raw[b + 1]), capi.ptr(raw[b]), capi.ptr(raw[b - 1]),
                                                    capi.ptr(addr), capi.stream_ptr()))
        n = _compact(capi, capi.LIB.ssrlcv_hip_compact_addresses, addr, w * h)
        kp = capi.dev_bytes(32 * max(n, 1))
        capi.check(capi.LIB.ssrlcv_hip_fill_extrema(u32(n), u32(w), u32(h), ci(o), ci(b), f32(float(sig[b])), capi.ptr(addr), capi.ptr(raw[b]),
                                                    capi.ptr(kp), capi.stream_ptr()))
        out.append(capi.to_host(kp, H.SSKEYPOINT, n))
    want, _ = _octave_slice(stages[0], o)
    got = np.concatenate(out)
    assert len(want) > 0
    _same(got, want)
    assert (got["discard"] == 0).all()


@pytest.mark.parametrize("o", [0, 1, 2])
def test_list_kernels_stage_by_stage(ctx, o):
    """flagNoise / refineLocation / flagEdges / checkKeyPoints + discardExtrema's remove_if, each applied to the oracle's
    list before the stage and compared with its list after it."""
    capi, s, stages = ctx
    w, h, pw, sig = s.octave_info(o)
    dogn = [_dev(l) for l in _levels(s, 2, o)]          # twice-normalised DoG levels (after :472)
    ptrs = torch.tensor([t.data_ptr() for t in dogn], dtype=torch.int64, device="cuda")

    def discard(kp_d, n):
        return _compact(capi, capi.LIB.ssrlcv_hip_compact_keypoints, kp_d, n)
    # 0 -> 1: removeNoise(0.8 x threshold)
    l0, _ = _octave_slice(stages[0], o)
    kp_d = _dev(l0)
    capi.check(capi.LIB.ssrlcv_hip_flag_noise(u32(len(l0)), capi.ptr(kp_d), f32(np.float32(NOISE * 0.8)), capi.stream_ptr()))
    n = discard(kp_d, len(l0))
    l1, _ = _octave_slice(stages[1], o)
    _same(capi.to_host(kp_d, H.SSKEYPOINT, n), l1)
    # 1 -> 2: refineLocation + discard; the stable sort by blur is the caller's (thrust::stable_sort upstream)
    kp_d = _dev(l1)
    sig_min = float(sig[0])
    mult = float(sig[1] / sig[0])
    capi.check(capi.LIB.ssrlcv_hip_refine_location(u32(len(l1)), u32(w), u32(h), f32(sig_min), f32(mult), u32(5), capi.ptr(ptrs), capi.ptr(kp_d),
                                                   capi.stream_ptr()))
    n = discard(kp_d, len(l1))
    got = capi.to_host(kp_d, H.SSKEYPOINT, n)
    got = got[np.argsort(got["blur"], kind="stable")]
    l2, idx2 = _octave_slice(stages[2], o)
    _same(got, l2)
    # 2 -> 3: removeNoise(threshold)
    kp_d = _dev(l2)
    capi.check(capi.LIB.ssrlcv_hip_flag_noise(u32(len(l2)), capi.ptr(kp_d), f32(np.float32(NOISE)), capi.stream_ptr()))
    n = discard(kp_d, len(l2))
    l3, idx3 = _octave_slice(stages[3], o)
    _same(capi.to_host(kp_d, H.SSKEYPOINT, n), l3)
    # 3 -> 4: removeEdges: flagEdges per blur segment on that blur's pixels (:287-306)
    kp_d = _dev(l3)
    for b in range(5):
        cnt = int((idx3[b + 1] if b < 4 else idx3[5]) - idx3[b])
        if cnt > 0:
            capi.check(capi.LIB.ssrlcv_hip_flag_edges(u32(cnt), u32(int(idx3[b])), u32(w), u32(h), capi.ptr(kp_d), capi.ptr(dogn[b]),
                                                      f32(EDGE), capi.stream_ptr()))
    n = discard(kp_d, len(l3))
    l4, idx4 = _octave_slice(stages[4], o)
    _same(capi.to_host(kp_d, H.SSKEYPOINT, n), l4)
    # 4 -> 5: checkKeyPoints over every segment (src/SIFT_FeatureFactory.cu:81-110)
    kp_d = _dev(l4)
    for b in range(5):
        cnt = int((idx4[b + 1] if b < 4 else idx4[5]) - idx4[b])
        if cnt > 0:
            capi.check(capi.LIB.ssrlcv_hip_check_keypoints(u32(cnt), u32(int(idx4[b])), u32(w), u32(h), f32(pw), f32(6.0), capi.ptr(kp_d),
                                                           capi.stream_ptr()))
    n = discard(kp_d, len(l4))
    l5, _ = _octave_slice(stages[5], o)
    _same(capi.to_host(kp_d, H.SSKEYPOINT, n), l5)
    assert len(l5) > 0 and (o != 0 or len(l1) < len(l0))


@pytest.mark.parametrize("o", [0, 1])
def test_orientation_and_descriptor_exports(ctx, oracle_lib, o):
    """calculatePixelGradients, computeThetas, the two thrust::remove calls, expandKeyPoints (src/FeatureFactory.cu:540-632)
    == the oracle's stage-6 list; fillDescriptors on it == the oracle's features, every byte."""
    capi, s, stages = ctx
    w, h, pw, sig = s.octave_info(o)
    dogn = _levels(s, 2, o)
    l5, idx5 = _octave_slice(stages[5], o)
    l6, idx6 = _octave_slice(stages[6], o)
    kp_d = _dev(l5)
    maxo = 2
    oriented, feats = [], []
    for b in range(5):
        cnt = int((idx5[b + 1] if b < 4 else idx5[5]) - idx5[b])
        if cnt <= 0:
            continue
        px = _dev(dogn[b])
        grad = capi.dev_bytes(8 * w * h)
        capi.check(capi.LIB.ssrlcv_hip_pixel_gradients(u32(w), u32(h), capi.ptr(px), capi.ptr(grad), capi.stream_ptr()))
        g = grad.cpu().numpy().view(np.float32).reshape(h, w, 2)
        lvl = dogn[b]
        assert np.array_equal(g[5, 7], np.array([lvl[5, 8] - lvl[5, 6], lvl[6, 7] - lvl[4, 7]], np.float32))
        assert np.array_equal(g[0, 0], np.array([lvl[0, 2] - lvl[0, 0], lvl[2, 0] - lvl[0, 0]], np.float32))   # border: the inner neighbour's stencil
        thetas = torch.zeros(cnt * maxo, dtype=torch.float32, device="cuda")
        nums = torch.zeros(cnt * maxo, dtype=torch.int32, device="cuda")
        capi.check(capi.LIB.ssrlcv_hip_compute_thetas(u32(cnt), u32(int(idx5[b])), u32(w), u32(h), f32(pw), f32(1.5), capi.ptr(kp_d),
                                                      capi.ptr(grad), capi.ptr(nums), u32(maxo), f32(0.8), capi.ptr(thetas), capi.stream_ptr()))
        nt = _compact(capi, capi.LIB.ssrlcv_hip_compact_thetas, thetas, cnt * maxo)
        nn = _compact(capi, capi.LIB.ssrlcv_hip_compact_addresses, nums, cnt * maxo)
        assert nt == nn
        out = capi.dev_bytes(32 * max(nn, 1))
        capi.check(capi.LIB.ssrlcv_hip_expand_keypoints(u32(nn), capi.ptr(kp_d), capi.ptr(out), capi.ptr(nums), capi.ptr(thetas),
                                                        capi.stream_ptr()))
        oriented.append(capi.to_host(out, H.SSKEYPOINT, nn))
        if nn:
            ft = torch.zeros(152 * nn, dtype=torch.uint8, device="cuda")
            capi.check(capi.LIB.ssrlcv_hip_fill_descriptors(u32(nn), u32(0), u32(w), u32(h), capi.ptr(ft), f32(pw), f32(6.0), capi.ptr(out),
                                                            capi.ptr(grad), capi.stream_ptr()))
            feats.append(capi.to_host(ft, H.FEATURE, nn))
    got = np.concatenate(oriented)
    assert len(l6) > 20
    _same(got, l6)
    # the oracle's features in list order: this octave's slice
    all_feats = s.features()
    start = int(sum(stages[6][1][k][5] for k in range(o)))
    want = all_feats[start:start + len(l6)]
    gf = np.concatenate(feats)
    assert np.array_equal(gf["values"], want["values"])
    for f in ("loc", "sigma", "theta"):
        assert np.array_equal(gf[f].view(np.uint32), want[f].view(np.uint32)), f


def test_in_place_compaction_over_many_tiles(ctx):
    capi, _, _ = ctx
    rng = np.random.default_rng(9)
    for n, p in ((1, 0.0), (1000, 0.5), (1 << 20, 0.97), (3000001, 0.3)):
        a = rng.integers(0, 1 << 30, n).astype(np.int32)
        a[rng.random(n) < p] = -1
        t = torch.from_numpy(a.copy()).cuda()
        cnt = _compact(capi, capi.LIB.ssrlcv_hip_compact_addresses, t, n)
        want = a[a != -1]
        assert cnt == len(want) and np.array_equal(t.cpu().numpy()[:cnt], want)
