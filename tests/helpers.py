"""Shared test helpers: POD dtypes (byte-compatible with include/ssrlcv_hip.h and
oracle/oracle.h), golden-fixture loaders and the ctypes handle on the CPU oracle.

The oracle is TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module's `oracle()`.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

KEYPOINT = np.dtype([("parentId", "<i4"), ("pad", "<i4"), ("loc", "<f4", (2,))])
MULTIMATCH = np.dtype([("numKeyPoints", "<u4"), ("index", "<i4")])
BUNDLE = np.dtype([("numLines", "<u4"), ("index", "<i4"), ("invalid", "u1"), ("pad", "u1", (3,))])
LINE = np.dtype([("vec", "<f4", (3,)), ("pnt", "<f4", (3,))])
FEATURE = np.dtype([("parent", "<i4"), ("pad", "<i4"), ("loc", "<f4", (2,)),
                    ("sigma", "<f4"), ("theta", "<f4"), ("values", "u1", (128,))])
SSKEYPOINT = np.dtype([("octave", "<i4"), ("blur", "<i4"), ("loc", "<f4", (2,)), ("intensity", "<f4"),
                       ("sigma", "<f4"), ("theta", "<f4"), ("discard", "u1"), ("pad", "u1", (3,))])
MATCH = np.dtype({"names": ["invalid", "kp0_parent", "kp0_loc", "kp1_parent", "kp1_loc"],
                  "formats": ["u1", "<i4", ("<f4", (2,)), "<i4", ("<f4", (2,))],
                  "offsets": [0, 8, 16, 24, 32], "itemsize": 40})
DMATCH = np.dtype({"names": ["invalid", "kp0_parent", "kp0_loc", "kp1_parent", "kp1_loc", "distance"],
                   "formats": ["u1", "<i4", ("<f4", (2,)), "<i4", ("<f4", (2,)), "<f4"],
                   "offsets": [0, 8, 16, 24, 32, 40], "itemsize": 48})
UINT2_PAIR = np.dtype([("a", "<u4", (2,)), ("b", "<u4", (2,))])
CAMERA = np.dtype({
    "names": ["cam_pos", "cam_rot", "fov", "foc", "dpix", "timeStamp", "ecef_offset", "no_rot", "size"],
    "formats": [("<f4", (3,)), ("<f4", (3,)), ("<f4", (2,)), "<f4", ("<f4", (2,)),
                "<i8", ("<f4", (3,)), "u1", ("<u4", (2,))],
    "offsets": [0, 12, 24, 32, 40, 48, 56, 68, 72], "itemsize": 80})
PUSHBROOM = np.dtype({
    "names": ["start_pos", "end_pos", "projection_center", "axis_radius", "roll", "altitude", "foc", "fov",
              "gsd", "dpix", "size"],
    "formats": [("<f4", (3,)), ("<f4", (3,)), ("<f4", (2,)), "<f4", "<f4", "<f4", "<f4", "<f4", "<f4",
                ("<f4", (2,)), ("<u4", (2,))],
    "offsets": [0, 12, 24, 32, 36, 40, 44, 48, 52, 56, 64], "itemsize": 72})

for _dt, _sz in ((KEYPOINT, 16), (MULTIMATCH, 8), (BUNDLE, 12), (LINE, 24), (FEATURE, 152), (SSKEYPOINT, 32),
                 (MATCH, 40), (DMATCH, 48), (UINT2_PAIR, 16), (CAMERA, 80), (PUSHBROOM, 72)):
    assert _dt.itemsize == _sz, (_dt, _sz)


def P(a):
    """numpy array -> void* (None passes NULL)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def load_view(view):
    """Returns dict with cameras (CAMERA[]), and per stage s: kp{s} (KEYPOINT[]), mm{s} (MULTIMATCH[]), points{s}."""
    d = np.load(os.path.join(GOLDEN, view + ".npz"))
    n = len(d["cam_foc"])
    cams = np.zeros(n, CAMERA)
    cams["cam_pos"] = d["cam_pos"]
    cams["cam_rot"] = d["cam_rot"]
    cams["fov"] = d["cam_fov"]
    cams["foc"] = d["cam_foc"]
    cams["dpix"] = d["cam_dpix"]
    cams["ecef_offset"] = d["cam_ecef_offset"]
    cams["size"] = d["cam_size"]
    out = {"cameras": cams}
    for s in (0, 1):
        kp = np.zeros(len(d["kp%d_parent" % s]), KEYPOINT)
        kp["parentId"] = d["kp%d_parent" % s]
        kp["loc"] = d["kp%d_loc" % s]
        mm = np.zeros(len(d["mm%d_num" % s]), MULTIMATCH)
        mm["numKeyPoints"] = d["mm%d_num" % s]
        mm["index"] = d["mm%d_index" % s]
        out["kp%d" % s] = kp
        out["mm%d" % s] = mm
        out["points%d" % s] = d["points%d" % s]
    if "points2" in d:
        out["points2"] = d["points2"]
    return out


def load_seed_features():
    d = np.load(os.path.join(GOLDEN, "seed_features.npz"))
    f = np.zeros(len(d["sigma"]), FEATURE)
    f["parent"] = d["parent"]
    f["loc"] = d["loc"]
    f["sigma"] = d["sigma"]
    f["theta"] = d["theta"]
    f["values"] = d["values"]
    return f, d["values_run2"]


def load_everest_pixels():
    d = np.load(os.path.join(GOLDEN, "everest_pixels.npz"))
    return [d["pixels_%d" % i] for i in range(3)]


_ORACLE = None


def cpu_budget():
    """CPUs this process may actually use: the affinity mask capped by the cgroup quota (a GPU box shows all 256 logical
    CPUs of its host but grants a share of them; an OpenMP team of 256 threads on a 16-CPU quota ran the oracle 50x
    slower, throttled and spinning)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def limit_openmp():
    """Size the OpenMP team of the oracle (and of anything else in the process) to cpu_budget(); must run before the
    OpenMP runtime starts, i.e. before the oracle library or torch is loaded.  -> the thread count"""
    n = int(os.environ.setdefault("OMP_NUM_THREADS", str(cpu_budget())))
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    return n


def dev_env(**switches):
    """Environment of a child process that loads the DEVELOPER build of the HIP library (ssrlcv_amd/_lib.py: the default
    is the release build, in which every SSRLCV_* switch is compiled out) with the given switches set."""
    env = dict(os.environ, SSRLCV_DEV_BUILD="1")
    env.pop("SSRLCV_HIP_LIB", None)
    env.update(switches)
    return env


# ---- the C++ host mirror's test drivers (ssrlcv_amd/host/_build) and the reference's on-disk formats -------------------------
HOST_MIRROR_BIN = os.path.join(ROOT, "ssrlcv_amd", "host", "_build", "host_mirror_test")
SHARDED_BIN = os.path.join(ROOT, "ssrlcv_amd", "host", "_build", "sharded_match_test")


def build_host_mirror():
    """-> path of host_mirror_test (built against the RELEASE flavour of the library, like a deployer's program)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "ssrlcv_amd", "csrc"), "release"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "ssrlcv_amd", "host")])
    return HOST_MIRROR_BIN


def host_typeinfo():
    """typeid names / hash codes of the checkpointable types as the mirror's compiler gives them (= the reference's .uty headers)."""
    out = subprocess.check_output([build_host_mirror(), "typeinfo"]).decode().split("\n")
    info = {}
    for line in out:
        if line.strip():
            label, name, h = line.split()
            info[label] = (name, int(h))
    return info


def write_uty(path, name, hash_code, state, data):
    """The reference's on-disk format (include/Unity.cuh:924-971)."""
    import struct
    raw = np.ascontiguousarray(data)
    with open(path, "wb") as f:
        f.write(name.encode() + b"\n")
        f.write(struct.pack("<Q", hash_code) + b"\n")
        f.write(struct.pack("<iQ", state, len(raw)) + b"\n")
        f.write(raw.tobytes())


def write_cpimg(path, image_id, size, camera):
    """An Image checkpoint as Image::checkpoint dumps it (src/Image.cu:274-303: the 240-byte object; the mirror reads its POD
    members at their offsets, host/Image.hpp): id @32, size @40, colorDepth @48, Camera (80 B) @56, isPushbroom @208."""
    import struct
    raw = bytearray(240)
    struct.pack_into("<i", raw, 32, int(image_id))
    struct.pack_into("<II", raw, 40, int(size[0]), int(size[1]))
    struct.pack_into("<I", raw, 48, 1)
    cam = np.ascontiguousarray(camera).view(np.uint8).reshape(-1)[:80]
    raw[56:136] = cam.tobytes()
    open(path, "wb").write(bytes(raw))


def oracle():
    """Build (if needed) and load oracle/_build/libssrlcv_oracle.so."""
    global _ORACLE
    if _ORACLE is None:
        limit_openmp()
        so = os.path.join(ROOT, "oracle", "_build", "libssrlcv_oracle.so")
        srcs = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle"))
                if f.endswith((".c", ".h"))]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
        lib = ctypes.CDLL(so)
        for name in ("oracle_two_view_triangulate", "oracle_n_view_triangulate", "oracle_ba_eval",
                     "oracle_dist_protocol"):
            getattr(lib, name).restype = ctypes.c_float
        lib.oracle_sift_create.restype = ctypes.c_void_p
        _ORACLE = lib
    return _ORACLE


def oracle_sift(lib, pixels, max_orientations=2, orientation_threshold=0.8, ori_width=1.5, desc_width=6.0):
    """Run the oracle's generateFeatures restatement on a u8 image -> FEATURE[]"""
    img = np.ascontiguousarray(pixels, dtype=np.uint8)
    h, w = img.shape
    out = ctypes.c_void_p()
    lib.oracle_sift_generate.restype = ctypes.c_int
    n = lib.oracle_sift_generate(P(img), ctypes.c_uint32(w), ctypes.c_uint32(h), ctypes.c_uint32(max_orientations),
                                 ctypes.c_float(orientation_threshold), ctypes.c_float(ori_width),
                                 ctypes.c_float(desc_width), ctypes.byref(out))
    assert n >= 0
    f = np.ctypeslib.as_array(ctypes.cast(out, ctypes.POINTER(ctypes.c_uint8)), shape=(n * 152,)).view(FEATURE).copy()
    lib.oracle_free(out)
    return f


def oracle_seed_distances(lib, query, seed):
    sd = np.zeros(len(query), np.float32)
    lib.oracle_seed_distances(ctypes.c_uint32(len(query)), P(query), ctypes.c_uint32(len(seed)), P(seed), P(sd))
    return sd


def oracle_projection(lib, cam):
    p4 = np.zeros((3, 4), np.float32)
    lib.oracle_projection_matrix(P(cam), P(p4))
    return p4


def oracle_match_dmatch(lib, mode, qid, q, tid, t, qcam, tproj, eps, delta, seed, rel, absolute):
    out = np.zeros(len(q), DMATCH)
    lib.oracle_match_dmatch(ctypes.c_int(mode), ctypes.c_uint32(qid), ctypes.c_uint32(len(q)), P(q),
                            ctypes.c_uint32(tid), ctypes.c_uint32(len(t)), P(t), P(qcam), P(tproj),
                            ctypes.c_float(eps), ctypes.c_float(delta), P(seed), ctypes.c_float(rel),
                            ctypes.c_float(absolute), P(out))
    return out


def oracle_match_match(lib, mode, qid, q, tid, t, qcam, tproj, eps, delta, seed, rel, absolute):
    out = np.zeros(len(q), MATCH)
    lib.oracle_match_match(ctypes.c_int(mode), ctypes.c_uint32(qid), ctypes.c_uint32(len(q)), P(q), ctypes.c_uint32(tid),
                           ctypes.c_uint32(len(t)), P(t), P(qcam) if qcam is not None else None,
                           P(tproj) if tproj is not None else None, ctypes.c_float(eps), ctypes.c_float(delta),
                           P(seed) if seed is not None else None, ctypes.c_float(rel), ctypes.c_float(absolute), P(out))
    return out


def oracle_match_pairs(lib, mode, qid, q, tid, t, qcam, tproj, eps, delta, seed, rel, absolute):
    out = np.zeros(len(q), UINT2_PAIR)
    lib.oracle_match_pairs(ctypes.c_int(mode), ctypes.c_uint32(qid), ctypes.c_uint32(len(q)), P(q),
                           ctypes.c_uint32(tid), ctypes.c_uint32(len(t)), P(t), P(qcam), P(tproj),
                           ctypes.c_float(eps), ctypes.c_float(delta), P(seed), ctypes.c_float(rel),
                           ctypes.c_float(absolute), P(out))
    return out


def oracle_merge(lib, num_features, pair_lists):
    counts = np.array([len(p) for p in pair_lists], np.uint32)
    allp = np.ascontiguousarray(np.concatenate(pair_lists))
    nf = np.array(num_features, np.uint32)
    mm_p, mem_p, nmem = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint32()
    lib.oracle_exhaustive_merge.restype = ctypes.c_int
    n = lib.oracle_exhaustive_merge(ctypes.c_uint32(len(nf)), P(nf), ctypes.c_uint32(len(counts)), P(counts), P(allp),
                                    ctypes.byref(mm_p), ctypes.byref(mem_p), ctypes.byref(nmem))
    mm = np.ctypeslib.as_array(ctypes.cast(mm_p, ctypes.POINTER(ctypes.c_uint8)), shape=(n * 8,)).view(MULTIMATCH).copy()
    mem = np.ctypeslib.as_array(ctypes.cast(mem_p, ctypes.POINTER(ctypes.c_uint32)), shape=(nmem.value, 2)).copy()
    lib.oracle_free(mm_p)
    lib.oracle_free(mem_p)
    return mm, mem


def oracle_pose_terms(lib, matches, pose6, qcam, tcam):
    pose = np.asarray(pose6, np.float32).copy()
    jtj, jtf, cost = np.zeros(36, np.float32), np.zeros(6, np.float32), ctypes.c_float(0)
    lib.oracle_pose_lm_terms(P(matches), ctypes.c_uint32(len(matches)), P(pose), P(qcam), P(tcam), P(jtj), P(jtf),
                             ctypes.byref(cost))
    return jtj.reshape(6, 6), jtf, float(cost.value)


def oracle_pose_cost(lib, matches, pose6, qcam, tcam):
    pose = np.asarray(pose6, np.float32).copy()
    lib.oracle_pose_cost.restype = ctypes.c_float
    return float(lib.oracle_pose_cost(P(matches), ctypes.c_uint32(len(matches)), P(pose), P(qcam), P(tcam)))


def relative_pose(cams):
    """Starting pose of PoseEstimator::LM_optimize / doPoseEstimation (src/Pipeline.cu:105-121, PoseEstimator.cu:323-329):
    rotation of camera 1 relative to camera 0, position in the frame of camera 0 in units of 1000 km."""
    def rot(a):
        x, y, z = [float(v) for v in a]
        cx, sx, cy, sy, cz, sz = np.cos(x), np.sin(x), np.cos(y), np.sin(y), np.cos(z), np.sin(z)
        return np.array([[cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx],
                         [sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx],
                         [-sy, cy * sx, cy * cx]])
    R0, R1 = rot(cams["cam_rot"][0]), rot(cams["cam_rot"][1])
    rel = R0.T @ R1
    x = np.arctan2(rel[2, 1], rel[2, 2])
    y = np.arctan2(-rel[2, 0], rel[2, 2] / np.cos(x))
    z = np.arctan2(rel[1, 0], rel[0, 0])
    pos = R0.T @ (cams["cam_pos"][1].astype(np.float64) - cams["cam_pos"][0].astype(np.float64))
    return np.array([x, y, z, pos[0] / 1000.0, pos[1] / 1000.0, pos[2] / 1000.0], np.float32)


def matches_from_matchset(kp):
    """2-view MatchSet key points (pairs) -> MATCH[]"""
    m = np.zeros(len(kp) // 2, MATCH)
    m["kp0_parent"], m["kp0_loc"] = kp["parentId"][0::2], kp["loc"][0::2]
    m["kp1_parent"], m["kp1_loc"] = kp["parentId"][1::2], kp["loc"][1::2]
    return m


def oracle_bundles(lib, mm, kp, cams):
    bundles = np.zeros(len(mm), BUNDLE)
    lines = np.zeros(len(kp), LINE)
    cams = cams.copy()
    lib.oracle_generate_bundles(ctypes.c_uint32(len(mm)), P(mm), P(kp), P(cams), P(bundles), P(lines))
    return bundles, lines, cams


def oracle_triangulate(lib, nview, bundles, lines, want_errors=False, cutoff=None):
    n = len(bundles)
    pts = np.zeros((n, 3), np.float32)
    errs = np.zeros(n, np.float32) if want_errors else None
    cut = np.array([cutoff], np.float32) if cutoff is not None else None
    f = lib.oracle_n_view_triangulate if nview else lib.oracle_two_view_triangulate
    total = f(ctypes.c_uint32(n), P(lines), P(bundles), P(pts), P(errs), P(cut))
    return pts, errs, total


def oracle_filter(lib, mm, kp, cams, kind, cutoff=None, sigma=None, sample_size=None):
    """linearCutoffFilter (kind "linear", src/PointCloudFactory.cu:3500-3644) or deterministicStatisticalFilter (kind
    "statistical", :3070-3275) on a MatchSet, restated on the oracle's pieces -> (MultiMatch, KeyPoint) after the filter."""
    nview = len(cams) > 2
    bundles, lines, _ = oracle_bundles(lib, mm, kp, cams)
    lib.oracle_sample_cutoff.restype = ctypes.c_float
    if kind == "linear":
        if cutoff < 0.0:
            return mm, kp
        oracle_triangulate(lib, nview, bundles, lines, want_errors=True, cutoff=cutoff)
        only_if_bad = True
    else:
        if sample_size > 1.0 or sample_size < 0.0:
            return mm, kp
        jump = int(1 / sample_size)
        _, errs, _ = oracle_triangulate(lib, nview, bundles, lines, want_errors=True, cutoff=0.0)
        cut = lib.oracle_sample_cutoff(P(errs), ctypes.c_uint32(len(errs)), ctypes.c_uint32(jump), ctypes.c_float(sigma))
        oracle_triangulate(lib, nview, bundles, lines, want_errors=True, cutoff=cut)
        only_if_bad = nview
    bad = int((bundles["invalid"] != 0).sum())
    bad_lines = int(bundles["numLines"][bundles["invalid"] != 0].sum())
    if only_if_bad and not bad:
        return mm, kp
    if len(mm) - bad == 0 or (nview and len(kp) - bad_lines == 0):
        return mm, kp      # "filtering is too aggressive": upstream leaves the MatchSet alone
    mm_out, kp_out = np.zeros(len(mm), MULTIMATCH), np.zeros(len(kp), KEYPOINT)
    counts = np.zeros(3, np.uint32)
    lib.oracle_filter_matchset(ctypes.c_uint32(len(mm)), P(bundles), P(kp), P(mm_out), P(kp_out), P(counts))
    return mm_out[:counts[0]].copy(), kp_out[:counts[1]].copy()


class OracleSift:
    """Staged access to the oracle's scale space for kernel-level parity tests."""

    def __init__(self, lib, pixels):
        self.lib = lib
        img = np.ascontiguousarray(pixels, dtype=np.uint8)
        self.h, self.w = img.shape
        lib.oracle_sift_create.restype = ctypes.c_void_p
        self.handle = ctypes.c_void_p(lib.oracle_sift_create(P(img), ctypes.c_uint32(self.w), ctypes.c_uint32(self.h)))
        assert self.handle.value

    def close(self):
        if self.handle:
            self.lib.oracle_sift_destroy(self.handle)
            self.handle = None

    def octave_info(self, o):
        w, h, pw = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_float()
        sig = np.zeros(6, np.float32)
        self.lib.oracle_sift_octave_info(self.handle, ctypes.c_int(o), ctypes.byref(w), ctypes.byref(h),
                                         ctypes.byref(pw), P(sig))
        return w.value, h.value, pw.value, sig

    def level(self, kind, o, b):
        """kind 0 normalised gaussian, 1 raw DoG, 2 twice-normalised DoG"""
        w, h, _, _ = self.octave_info(o)
        out = np.zeros((h, w), np.float32)
        self.lib.oracle_sift_level(self.handle, ctypes.c_int(kind), ctypes.c_int(o), ctypes.c_int(b), P(out), None, None)
        return out

    def minmax(self, kind, o, b):
        mn, mx = ctypes.c_float(), ctypes.c_float()
        self.lib.oracle_sift_minmax(self.handle, ctypes.c_int(kind), ctypes.c_int(o), ctypes.c_int(b),
                                    ctypes.byref(mn), ctypes.byref(mx))
        return mn.value, mx.value

    def keypoints(self, stage):
        out = ctypes.c_void_p()
        idx = np.zeros((4, 6), np.int32)
        self.lib.oracle_sift_keypoints.restype = ctypes.c_int
        n = self.lib.oracle_sift_keypoints(self.handle, ctypes.c_int(stage), ctypes.byref(out), P(idx))
        kps = np.ctypeslib.as_array(ctypes.cast(out, ctypes.POINTER(ctypes.c_uint8)),
                                    shape=(max(n, 1) * 32,)).view(SSKEYPOINT)[:n].copy()
        self.lib.oracle_free(out)
        return kps, idx

    def features(self, max_orientations=2, thr=0.8, ow=1.5, dw=6.0):
        out = ctypes.c_void_p()
        self.lib.oracle_sift_features.restype = ctypes.c_int
        n = self.lib.oracle_sift_features(self.handle, ctypes.c_uint32(max_orientations), ctypes.c_float(thr),
                                          ctypes.c_float(ow), ctypes.c_float(dw), ctypes.byref(out))
        f = np.ctypeslib.as_array(ctypes.cast(out, ctypes.POINTER(ctypes.c_uint8)),
                                  shape=(max(n, 1) * 152,)).view(FEATURE)[:n].copy()
        self.lib.oracle_free(out)
        return f


def oracle_gauss_kernel(lib, sigma, pixel_width):
    w = np.zeros(257, np.float32)
    lib.oracle_gauss_kernel.restype = ctypes.c_int
    taps = lib.oracle_gauss_kernel(ctypes.c_float(sigma), ctypes.c_float(pixel_width), P(w))
    return taps, w[:taps].copy()


def synthetic_image(w, h, seed=0, blobs=None):
    """Deterministic textured test image (smooth noise + gaussian blobs), u8, mean ~128: gives stable DoG extrema.
    Pure numpy so the same image exists on the GPU box."""
    rng = np.random.default_rng(0x53524C43 + seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w), np.float32)
    for octv in range(5):
        gw, gh = max(2, (w >> (5 - octv)) + 2), max(2, (h >> (5 - octv)) + 2)
        grid = rng.standard_normal((gh, gw)).astype(np.float32)
        gx = xx * (gw - 1.001) / w
        gy = yy * (gh - 1.001) / h
        x0, y0 = gx.astype(int), gy.astype(int)
        fx, fy = gx - x0, gy - y0
        fx = fx * fx * (3 - 2 * fx)
        fy = fy * fy * (3 - 2 * fy)
        v = (grid[y0, x0] * (1 - fx) * (1 - fy) + grid[y0, x0 + 1] * fx * (1 - fy) +
             grid[y0 + 1, x0] * (1 - fx) * fy + grid[y0 + 1, x0 + 1] * fx * fy)
        img += v * (24.0 * 0.6 ** octv)
    nb = blobs if blobs is not None else max(16, (w * h) // 1024)
    bx = rng.uniform(0, w, nb)
    by = rng.uniform(0, h, nb)
    bs = rng.uniform(1.5, 6.0, nb)
    ba = rng.uniform(-40, 40, nb)
    for i in range(nb):
        r = int(4 * bs[i]) + 1
        xa, xb = max(0, int(bx[i]) - r), min(w, int(bx[i]) + r + 1)
        ya, yb = max(0, int(by[i]) - r), min(h, int(by[i]) + r + 1)
        if xa >= xb or ya >= yb:
            continue
        sub = np.exp(-(((xx[ya:yb, xa:xb] - bx[i]) ** 2 + (yy[ya:yb, xa:xb] - by[i]) ** 2) / (2 * bs[i] ** 2)))
        img[ya:yb, xa:xb] += ba[i] * sub
    return np.clip(np.rint(img + 128.0), 0, 255).astype(np.uint8)


def bits(a):
    """float32 array -> its bit patterns (so that -0.0 / NaN payloads count, and == means bit-equal)."""
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_features_equal(gf, of):
    """Every field the reference writes is bit-equal (the 4 pad bytes after `parent` are never written upstream)."""
    assert len(gf) == len(of), (len(gf), len(of))
    assert np.array_equal(gf["parent"], of["parent"])
    for name in ("loc", "sigma", "theta"):
        ne = bits(gf[name]) != bits(of[name])
        assert not ne.any(), (name, int(ne.sum()), len(gf))
    nd = (gf["values"] != of["values"]).any(1)
    assert not nd.any(), ("descriptor bytes", int(nd.sum()), len(gf))
