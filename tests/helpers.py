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


def oracle():
    """Build (if needed) and load oracle/_build/libssrlcv_oracle.so."""
    global _ORACLE
    if _ORACLE is None:
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
