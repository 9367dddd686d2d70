"""Thin Python plumbing over the C ABI (include/ssrlcv_hip.h) for tests and bench.py.

torch is used only for device memory and streams; every compute call goes through libssrlcv_hip.so with raw
device pointers.  Struct arrays travel as uint8 tensors (byte-compatible with include/ssrlcv_types.h).
"""
import ctypes

import numpy as np
import torch

from . import _lib

LIB = _lib.load()

c_u32, c_f32, c_vp, c_sz, c_int = ctypes.c_uint32, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int

OUT_DMATCH, OUT_UINT2_PAIR, OUT_MATCH = 0, 1, 2
_OUT_SIZE = {OUT_DMATCH: 48, OUT_UINT2_PAIR: 16, OUT_MATCH: 40}


class SsrlcvError(RuntimeError):
    pass


class MalformedPairList(SsrlcvError):
    """the device merge's input status word: an index out of range or a query matched twice in one pair"""


def check(rc):
    if rc != 0:
        raise SsrlcvError("ssrlcv_hip status %d: %s" % (rc, LIB.ssrlcv_hip_status_string(int(rc)).decode()))


def stream_ptr():
    return c_vp(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return c_vp(0)
    assert t.is_cuda and t.is_contiguous()
    return c_vp(t.data_ptr())


def to_dev(arr):
    """numpy (possibly structured) array -> uint8 CUDA tensor holding the same bytes."""
    a = np.ascontiguousarray(arr)
    return torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).cuda()


def to_host(t, dtype, count=None):
    a = t.detach().cpu().numpy().view(np.uint8).reshape(-1)
    dt = np.dtype(dtype)
    if count is not None:
        a = a[: count * dt.itemsize]
    return a.view(dt).copy()


def dev_bytes(n):
    return torch.empty(max(int(n), 1), dtype=torch.uint8, device="cuda")


class CameraStruct(ctypes.Structure):
    _fields_ = [("raw", ctypes.c_uint8 * 80)]


class MatchParams(ctypes.Structure):
    _fields_ = [("mode", c_int), ("queryImageID", c_u32), ("targetImageID", c_u32), ("epsilon", c_f32),
                ("delta", c_f32), ("relativeThreshold", c_f32), ("absoluteThreshold", c_f32),
                ("pad0", c_u32),  # ssrlcv_camera is 8-byte aligned
                ("queryCamera", ctypes.c_uint8 * 80), ("targetProjection", c_f32 * 12), ("fundamental", c_f32 * 9),
                ("pad1", c_u32 * 3)]  # the float4 member makes the struct 16-byte aligned


assert ctypes.sizeof(MatchParams) == 32 + 80 + 48 + 48, ctypes.sizeof(MatchParams)


class SiftParams(ctypes.Structure):
    _fields_ = [("maxOrientations", c_u32), ("orientationThreshold", c_f32), ("orientationContribWidth", c_f32),
                ("descriptorContribWidth", c_f32), ("maxKeyPointsPerOctave", c_u32)]


# ------------------------------------------------------------------ point cloud
def generate_bundles(matches_d, keypoints_d, num_bundles, cameras_d, num_cameras, num_keypoints):
    bundles = dev_bytes(num_bundles * 12)
    lines = dev_bytes(num_keypoints * 24)
    check(LIB.ssrlcv_hip_generate_bundles(ptr(matches_d), ptr(keypoints_d), c_u32(num_bundles), ptr(cameras_d),
                                          c_u32(num_cameras), ptr(bundles), ptr(lines), stream_ptr()))
    return bundles, lines


def generate_pushbroom_bundles(matches_d, keypoints_d, num_bundles, pushbrooms_d, num_cameras, num_keypoints):
    bundles = dev_bytes(num_bundles * 12)
    lines = dev_bytes(num_keypoints * 24)
    check(LIB.ssrlcv_hip_generate_pushbroom_bundles(ptr(matches_d), ptr(keypoints_d), c_u32(num_bundles),
                                                    ptr(pushbrooms_d), c_u32(num_cameras), ptr(bundles), ptr(lines),
                                                    stream_ptr()))
    return bundles, lines


def triangulate(lines_d, bundles_d, n, nview=False, want_points=True, want_errors=False, cutoff=None,
                no_error_variant=False):
    points = torch.zeros(max(n, 1) * 3, dtype=torch.float32, device="cuda") if want_points else None
    errors = torch.zeros(max(n, 1), dtype=torch.float32, device="cuda") if want_errors else None
    # cutoff: a float, or a 1-element float32 CUDA tensor (a cutoff computed on the device: error_sample_cutoff)
    cut = None if cutoff is None else (cutoff if torch.is_tensor(cutoff) else torch.tensor([cutoff], dtype=torch.float32, device="cuda"))
    esum = torch.zeros(1, dtype=torch.float32, device="cuda")
    if nview:
        check(LIB.ssrlcv_hip_triangulateN(ptr(lines_d), ptr(bundles_d), c_u32(n), ptr(points), ptr(errors), ptr(cut),
                                          ptr(esum), c_int(1 if no_error_variant else 0), stream_ptr()))
    else:
        check(LIB.ssrlcv_hip_triangulate2(ptr(lines_d), ptr(bundles_d), c_u32(n), ptr(points), ptr(errors), ptr(cut),
                                          ptr(esum), stream_ptr()))
    return points, errors, esum


def error_sample_cutoff(errors_d, n, sample_jump, sigma):
    """deterministicStatisticalFilter's cutoff (sigma x std of every sample_jump-th error, sequential float sums) on the
    device -> 1-element float32 CUDA tensor."""
    cut = torch.zeros(1, dtype=torch.float32, device="cuda")
    check(LIB.ssrlcv_hip_error_sample_cutoff(ptr(errors_d), c_u32(n), c_u32(sample_jump), c_f32(sigma), ptr(cut), stream_ptr()))
    return cut


def filter_matchset(bundles_d, keypoints_d, num_bundles, num_keypoints):
    """The MatchSet without the bundles flagged invalid -> (MultiMatch bytes, KeyPoint bytes, counts tensor {bundles kept,
    key points kept, key points in}); asynchronous."""
    LIB.ssrlcv_hip_filter_workspace_bytes.restype = ctypes.c_size_t
    ws = dev_bytes(int(LIB.ssrlcv_hip_filter_workspace_bytes(c_u32(num_bundles))))
    mm = dev_bytes(8 * max(num_bundles, 1))
    kp = dev_bytes(16 * max(num_keypoints, 1))
    counts = torch.zeros(3, dtype=torch.int32, device="cuda")
    check(LIB.ssrlcv_hip_filter_matchset(ptr(bundles_d), ptr(keypoints_d), c_u32(num_bundles), ptr(mm), ptr(kp), ptr(counts),
                                         ptr(ws), c_sz(ws.numel()), stream_ptr()))
    return mm, kp, counts


def select_pair_bundles(matches_d, keypoints_d, num_matches, num_keypoints, image_a, image_b):
    """The two-view bundles of image pair (a, b) of a MatchSet on the device as a two-camera MatchSet
    (ssrlcv_hip_select_pair_bundles) -> (MultiMatch bytes, KeyPoint bytes, count tensor); asynchronous."""
    ws = dev_bytes(int(LIB.ssrlcv_hip_select_pair_workspace_bytes(c_u32(num_matches))))
    mm = dev_bytes(8 * max(num_matches, 1))
    kp = dev_bytes(32 * max(num_matches, 1))
    count = torch.zeros(1, dtype=torch.int32, device="cuda")
    check(LIB.ssrlcv_hip_select_pair_bundles(ptr(matches_d), ptr(keypoints_d), c_u32(num_matches), c_u32(num_keypoints),
                                             c_int(image_a), c_int(image_b), ptr(mm), ptr(kp), ptr(count), ptr(ws), c_sz(ws.numel()),
                                             stream_ptr()))
    return mm, kp, count


def ba_sweep2(matches_d, keypoints_d, num_bundles, cameras_d, num_cameras, params_d, K):
    sums = torch.zeros(K, dtype=torch.float32, device="cuda")
    check(LIB.ssrlcv_hip_ba_sweep2(ptr(matches_d), ptr(keypoints_d), c_u32(num_bundles), ptr(cameras_d),
                                   c_u32(num_cameras), ptr(params_d), c_u32(K), ptr(sums), c_vp(0), c_sz(0),
                                   stream_ptr()))
    return sums


# ------------------------------------------------------------------ pose refinement
def _host_bytes(a, n):
    return np.ascontiguousarray(a).view(np.uint8).reshape(-1)[:n].copy()


def pose_lm_terms(matches_d, n, pose6, query_cam_np, target_cam_np):
    """-> (JTJ[6,6] with JTJ[j, i] = out[i + 6 j], JTf[6], cost) of PoseEstimator::LM_iteration at `pose6`
    (roll, pitch, yaw, x, y, z)."""
    pose = np.asarray(pose6, np.float32).copy()
    q, t = _host_bytes(query_cam_np, 80), _host_bytes(target_cam_np, 80)
    out = torch.empty(43, dtype=torch.float32, device="cuda")
    check(LIB.ssrlcv_hip_pose_lm_terms(ptr(matches_d), c_u32(n), pose.ctypes.data_as(c_vp), q.ctypes.data_as(c_vp),
                                       t.ctypes.data_as(c_vp), ptr(out), stream_ptr()))
    o = out.cpu().numpy()
    return o[:36].reshape(6, 6).copy(), o[36:42].copy(), float(o[42])


def pose_cost(matches_d, n, pose6, query_cam_np, target_cam_np):
    pose = np.asarray(pose6, np.float32).copy()
    q, t = _host_bytes(query_cam_np, 80), _host_bytes(target_cam_np, 80)
    out = torch.empty(1, dtype=torch.float32, device="cuda")
    check(LIB.ssrlcv_hip_pose_cost(ptr(matches_d), c_u32(n), pose.ctypes.data_as(c_vp), q.ctypes.data_as(c_vp),
                                   t.ctypes.data_as(c_vp), ptr(out), stream_ptr()))
    return float(out.item())


# ------------------------------------------------------------------ matching
def projection_matrix(camera_np):
    cam = np.ascontiguousarray(camera_np).view(np.uint8).reshape(-1)[:80].copy()
    out = np.zeros(12, np.float32)
    LIB.ssrlcv_projection_matrix_host(cam.ctypes.data_as(c_vp), out.ctypes.data_as(c_vp))
    return out.reshape(3, 4)


def match_workspace(nq, nt):
    return dev_bytes(LIB.ssrlcv_hip_match_workspace_bytes(c_u32(nq), c_u32(nt)))


def seed_distances(query_d, nq, seed_d, ns, workspace=None):
    ws = workspace if workspace is not None else match_workspace(nq, ns)
    out = torch.empty(max(nq, 1), dtype=torch.float32, device="cuda")
    check(LIB.ssrlcv_hip_seed_distances_u8x128(ptr(query_d), c_u32(nq), ptr(seed_d), c_u32(ns), ptr(out), ptr(ws),
                                               c_sz(ws.numel()), stream_ptr()))
    return out


def make_match_params(mode, query_id, target_id, epsilon=0.0, delta=0.0, rel=0.0, absolute=0.0, query_camera=None,
                      target_projection=None, fundamental=None):
    p = MatchParams()
    p.mode, p.queryImageID, p.targetImageID = mode, query_id, target_id
    p.epsilon, p.delta, p.relativeThreshold, p.absoluteThreshold = epsilon, delta, rel, absolute
    if query_camera is not None:
        raw = np.ascontiguousarray(query_camera).view(np.uint8).reshape(-1)[:80]
        ctypes.memmove(p.queryCamera, raw.ctypes.data, 80)
    if target_projection is not None:
        tp = np.ascontiguousarray(target_projection, dtype=np.float32).reshape(-1)
        ctypes.memmove(p.targetProjection, tp.ctypes.data, 48)
    if fundamental is not None:
        f9 = np.ascontiguousarray(fundamental, dtype=np.float32).reshape(-1)
        ctypes.memmove(p.fundamental, f9.ctypes.data, 36)
    return p


MATCH_ARITH_I8, MATCH_ARITH_F16 = 0, 1


def set_match_arithmetic(arithmetic):
    """int8 MFMA (default) or fp16 MFMA behind every matcher call of this process: both exact, same outputs."""
    check(LIB.ssrlcv_hip_set_match_arithmetic(c_int(arithmetic)))


def get_match_arithmetic():
    return int(LIB.ssrlcv_hip_get_match_arithmetic())


def match(query_d, nq, target_d, nt, params, out_kind=OUT_DMATCH, seed_d=None, workspace=None, out=None):
    ws = workspace if workspace is not None else match_workspace(nq, nt)
    if out is None:
        out = dev_bytes(nq * _OUT_SIZE[out_kind])
    check(LIB.ssrlcv_hip_match_u8x128(ptr(query_d), c_u32(nq), ptr(target_d), c_u32(nt), ptr(seed_d),
                                      ctypes.byref(params), c_int(out_kind), ptr(out), ptr(ws), c_sz(ws.numel()),
                                      stream_ptr()))
    return out


def compact_matches(out_kind, matches_d, n, workspace):
    cnt = c_u32(0)
    check(LIB.ssrlcv_hip_compact_matches(c_int(out_kind), ptr(matches_d), c_u32(n), ctypes.byref(cnt), ptr(workspace),
                                         c_sz(workspace.numel()), stream_ptr()))
    return cnt.value


def compact_matches_async(out_kind, matches_d, n, workspace, count_d):
    """Compaction without the host round trip: count_d (1-element int32 CUDA tensor / slice) receives the survivors."""
    check(LIB.ssrlcv_hip_compact_matches_async(c_int(out_kind), ptr(matches_d), c_u32(n), ptr(count_d), ptr(workspace),
                                               c_sz(workspace.numel()), stream_ptr()))


def sort_keys(keys_d, n):
    """perm (int32 CUDA tensor of n uint32) ordering 0 .. n-1 by ascending (keys[i], i) -- the sort of the band-culled modes"""
    ws = dev_bytes(int(LIB.ssrlcv_hip_sort_workspace_bytes(c_u32(n))))
    perm = torch.empty(max(n, 1), dtype=torch.int32, device="cuda")
    check(LIB.ssrlcv_hip_sort_keys_u32(ptr(keys_d), c_u32(n), ptr(perm), ptr(ws), c_sz(ws.numel()), stream_ptr()))
    return perm[:n]


def keypoints_from_members(members_d, n, feature_tensors):
    """KeyPoint{image, location} of every {image, feature} member, gathered on the device -> uint8 tensor (16 B each)."""
    V = len(feature_tensors)
    assert all(t is None or t.is_cuda for t in feature_tensors), "feature arrays must live on the device"
    ptrs = (c_vp * V)(*[t.data_ptr() if t is not None and t.numel() else 0 for t in feature_tensors])
    counts = (c_u32 * V)(*[(t.numel() // 152) if t is not None else 0 for t in feature_tensors])
    out = dev_bytes(16 * n)
    check(LIB.ssrlcv_hip_keypoints_from_members(ptr(members_d), c_u32(n), ptrs, counts, c_u32(V), ptr(out), stream_ptr()))
    return out


def merge_matches_device(num_features, pair_counts, pairs_d, workspace=None):
    """Device merge of the validated uint2_pair arrays of all image pairs (concatenated in pair order, on the device) ->
    (MultiMatch bytes, member bytes, numMatches, numMembers, rounds).  One small D2H copy of the two counts."""
    V = len(num_features)
    nf = (c_u32 * V)(*[int(x) for x in num_features])
    pc = (c_u32 * max(len(pair_counts), 1))(*[int(x) for x in pair_counts])
    total = int(sum(int(x) for x in pair_counts))
    LIB.ssrlcv_hip_merge_workspace_bytes.restype = ctypes.c_size_t
    need = int(LIB.ssrlcv_hip_merge_workspace_bytes(c_u32(V), nf, c_u32(total)))
    if need == 0:
        raise ValueError("ssrlcv_hip_merge_matches: 2..32 images")
    if workspace is None or workspace.numel() < need:
        workspace = dev_bytes(need)
    mm = dev_bytes(8 * max(total, 1))
    mem = dev_bytes(8 * 2 * max(total, 1))
    counts = torch.zeros(4, dtype=torch.int32, device="cuda")
    check(LIB.ssrlcv_hip_merge_matches(c_u32(V), nf, c_u32(len(pair_counts)), pc, ptr(pairs_d) if total else None, ptr(workspace),
                                       c_sz(workspace.numel()), ptr(mm), ptr(mem), ptr(counts), stream_ptr()))
    n_mm, n_mem, bad, rounds = [int(x) for x in counts.cpu().tolist()]  # the one synchronisation of the (asynchronous) call
    if bad & 4:  # (bit 2: a block of the persistent walk was not resident and its grid barrier gave up, csrc/merge.hip)
        raise SsrlcvError("ssrlcv_hip_merge_matches: the persistent walk could not keep all its blocks resident (grid barrier timed out)")
    if bad:
        raise MalformedPairList("ssrlcv_hip_merge_matches: malformed pair list (status word %d)" % bad)
    return mm[: 8 * n_mm], mem[: 8 * n_mem], n_mm, n_mem, rounds, workspace


def matchset_from_matches(in_kind, matches_d, n, want_max=False):
    """Device M7: (KeyPoint[2n] bytes, MultiMatch[n] bytes, max distance or None) from a validated DMatch / Match array."""
    kp = dev_bytes(32 * n)
    mm = dev_bytes(8 * n)
    mx = torch.full((1,), -1.0, dtype=torch.float32, device="cuda") if want_max else None
    check(LIB.ssrlcv_hip_matchset_from_matches(c_int(in_kind), ptr(matches_d), c_u32(n), ptr(kp), ptr(mm), ptr(mx),
                                               stream_ptr()))
    return kp, mm, (float(mx.item()) if want_max else None)


# ------------------------------------------------------------------ SIFT kernel-level
def gauss_kernel(sigma, pixel_width):
    w = np.zeros(129, np.float32)
    taps = LIB.ssrlcv_gauss_kernel_host(c_f32(sigma), c_f32(pixel_width), w.ctypes.data_as(c_vp))
    return taps, w[:max(taps, 0)].copy()


def convert_to_bw(color_d, depth, n):
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    check(LIB.ssrlcv_hip_convert_to_bw(ptr(color_d), c_u32(depth), ptr(out), c_sz(n), stream_ptr()))
    return out


def upsample2x_u8(img_d, w, h):
    out = torch.empty(4 * w * h, dtype=torch.float32, device="cuda")
    check(LIB.ssrlcv_hip_upsample2x_u8(ptr(img_d), c_u32(w), c_u32(h), ptr(out), stream_ptr()))
    return out


def upsample2x(img_d, w, h):
    out = torch.empty(4 * w * h, dtype=torch.float32, device="cuda")
    check(LIB.ssrlcv_hip_upsample2x(ptr(img_d), c_u32(w), c_u32(h), ptr(out), stream_ptr()))
    return out


def u8_to_f32(img_d, n):
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    check(LIB.ssrlcv_hip_u8_to_f32(ptr(img_d), ptr(out), c_sz(n), stream_ptr()))
    return out


def bin2x(img_d, w, h):
    out = torch.empty((w // 2) * (h // 2), dtype=torch.float32, device="cuda")
    check(LIB.ssrlcv_hip_bin2x(ptr(img_d), c_u32(w), c_u32(h), ptr(out), stream_ptr()))
    return out


def gauss_sep_conv(img_d, w, h, weights, want_minmax=True):
    out = torch.empty(w * h, dtype=torch.float32, device="cuda")
    mm = torch.tensor([3.4028234663852886e38, -3.4028234663852886e38], dtype=torch.float32, device="cuda") \
        if want_minmax else None
    wh = np.ascontiguousarray(weights, dtype=np.float32)
    check(LIB.ssrlcv_hip_gauss_sep_conv(ptr(img_d), ptr(out), c_vp(0), c_u32(w), c_u32(h), c_int(len(wh)),
                                        wh.ctypes.data_as(c_vp), ptr(mm), stream_ptr()))
    return out, mm


def minmax(img_d, n):
    mm = torch.empty(2, dtype=torch.float32, device="cuda")
    check(LIB.ssrlcv_hip_minmax(ptr(img_d), c_sz(n), ptr(mm), stream_ptr()))
    return mm


def normalize_(img_d, n, mm_d):
    check(LIB.ssrlcv_hip_normalize(ptr(img_d), c_sz(n), ptr(mm_d), stream_ptr()))


def math_eval(fn, a, b=None):
    """Test hook: element-wise device elementary function (0 expf, 1 atan2f(a,b), 2 sinf, 3 cosf, 4 tanf, 5 powf(a,b))."""
    a_d = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    b_d = None if b is None else torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32)).cuda()
    out = torch.empty_like(a_d)
    check(LIB.ssrlcv_hip_math_eval(c_int(fn), ptr(a_d), ptr(b_d), ptr(out), c_sz(a_d.numel()), stream_ptr()))
    return out.cpu().numpy()


# ------------------------------------------------------------------ SIFT pipeline
class SiftPlan:
    """Owns an ssrlcv_sift_plan and (optionally) the workspace tensor for one W x H image slot."""

    def __init__(self, w, h, max_orientations=2, orientation_threshold=0.8, orientation_contrib_width=1.5,
                 descriptor_contrib_width=6.0, max_keypoints_per_octave=0, alloc=True):
        self.w, self.h = w, h
        p = SiftParams(max_orientations, orientation_threshold, orientation_contrib_width, descriptor_contrib_width,
                       max_keypoints_per_octave)
        self.handle = c_vp()
        check(LIB.ssrlcv_sift_plan_create(c_u32(w), c_u32(h), ctypes.byref(p), ctypes.byref(self.handle)))
        self.workspace_bytes = LIB.ssrlcv_sift_plan_workspace_bytes(self.handle)
        self.max_features = LIB.ssrlcv_sift_plan_max_features(self.handle)
        self.workspace = dev_bytes(self.workspace_bytes) if alloc else None
        self.features = dev_bytes(self.max_features * 152) if alloc else None
        self.num_features = torch.zeros(1, dtype=torch.int32, device="cuda") if alloc else None

    def __del__(self):
        if getattr(self, "handle", None) and LIB is not None:  # (module globals are gone at interpreter shutdown)
            LIB.ssrlcv_sift_plan_destroy(self.handle)
            self.handle = None

    def set_stop_stage(self, stage):
        LIB.ssrlcv_sift_plan_set_stop_stage(self.handle, c_int(stage))

    def set_stage_event(self, event):
        """A torch.cuda.Event (or None) that extract() records on the launching stream between the scale-space stage and the
        key-point stage (ssrlcv_sift_plan_set_stage_event): how a caller times the stages of the fused call."""
        handle = None
        if event is not None:
            event.record()           # torch creates the underlying hipEvent_t lazily, on the first record
            handle = event.cuda_event
        self._stage_event = event    # (kept alive as long as the plan refers to it)
        check(LIB.ssrlcv_sift_plan_set_stage_event(self.handle, c_vp(handle) if handle is not None else None))

    def build_dog(self, pixels_d):
        check(LIB.ssrlcv_hip_sift_build_dog(self.handle, ptr(pixels_d), ptr(self.workspace), stream_ptr()))

    def describe(self):
        check(LIB.ssrlcv_hip_sift_describe(self.handle, ptr(self.workspace), ptr(self.features),
                                           ptr(self.num_features), stream_ptr()))

    def stage(self, stage):
        """One reference launch site of the key-point stage (ssrlcv_hip_sift_stage), on the state the previous one left."""
        check(LIB.ssrlcv_hip_sift_stage(self.handle, ptr(self.workspace), c_int(stage), ptr(self.features),
                                        ptr(self.num_features), stream_ptr()))

    def extract(self, pixels_d):
        check(LIB.ssrlcv_hip_sift_extract(self.handle, ptr(pixels_d), ptr(self.workspace), ptr(self.features),
                                          ptr(self.num_features), stream_ptr()))

    def count(self):
        """Feature count of the last extract; raises when a key-point list outgrew its capacity (the list was truncated:
        not the reference's result -- re-run with a larger max_keypoints_per_octave)."""
        mask = c_u32(0)
        rc = LIB.ssrlcv_sift_plan_overflow(self.handle, ptr(self.workspace), ctypes.byref(mask), stream_ptr())
        if mask.value:
            raise SsrlcvError("key-point capacity exceeded in octaves %s" % [o for o in range(4) if mask.value >> o & 1])
        check(rc)
        return int(self.num_features.item())

    def overflow_mask(self):
        mask = c_u32(0)
        LIB.ssrlcv_sift_plan_overflow(self.handle, ptr(self.workspace), ctypes.byref(mask), stream_ptr())
        return mask.value

    def level(self, kind, octave, blur):
        """-> (numpy level copy, (min, max)); kind 0 = raw DoG, 1 = gaussian (last octave built)."""
        data, mm = c_vp(), c_vp()
        w, h = c_u32(), c_u32()
        check(LIB.ssrlcv_sift_plan_level(self.handle, ptr(self.workspace), c_int(kind), c_int(octave), c_int(blur),
                                         ctypes.byref(data), ctypes.byref(w), ctypes.byref(h), ctypes.byref(mm)))
        torch.cuda.synchronize()
        base = self.workspace.data_ptr()
        off = data.value - base
        n = w.value * h.value
        lvl = self.workspace[off: off + 4 * n].view(torch.float32).cpu().numpy().reshape(h.value, w.value).copy()
        moff = mm.value - base
        mmv = self.workspace[moff: moff + 8].view(torch.float32).cpu().numpy().copy()
        return lvl, (float(mmv[0]), float(mmv[1]))

    def keypoints(self, octave, dtype):
        """-> (SSKeyPoint numpy array of the octave, blur indices [6])"""
        lst, idx = c_vp(), c_vp()
        check(LIB.ssrlcv_sift_plan_keypoints(self.handle, ptr(self.workspace), c_int(octave), ctypes.byref(lst),
                                             ctypes.byref(idx)))
        torch.cuda.synchronize()
        base = self.workspace.data_ptr()
        ioff = idx.value - base
        state = self.workspace[ioff: ioff + 32].view(torch.int32).cpu().numpy().copy()
        n = int(state[5]) if int(state[6]) else 0
        loff = lst.value - base
        kps = to_host(self.workspace[loff: loff + 32 * max(n, 1)], dtype, n)
        return kps, state[:6].copy(), int(state[7])

    def features_host(self, dtype):
        n = self.count()
        return to_host(self.features[: 152 * max(n, 1)], dtype, n)
