"""The C++ host mirror of the reference API (ssrlcv_amd/host/*.hpp: Unity<T>, ptr::*, Image, SIFT_FeatureFactory,
MatchFactory<T>, PointCloudFactory) driven through tests/cpp/host_mirror_test.cpp.

CPU part: Unity state machine / exceptions / checkpoint format compatibility with the reference's .uty and .cpimg
files.  GPU part: the reference's stage flow (src/Pipeline.cu doFeatureGeneration -> doFeatureMatching ->
doTriangulation -> doBundleAdjust) through the class API on the everest fixtures, compared with the golden outputs.
"""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

import helpers as H

BIN = os.path.join(H.ROOT, "ssrlcv_amd", "host", "_build", "host_mirror_test")


def build_binary():
    subprocess.check_call(["make", "-s", "-C", os.path.join(H.ROOT, "ssrlcv_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(H.ROOT, "ssrlcv_amd", "host")])
    return BIN


def typeinfo():
    out = subprocess.check_output([build_binary(), "typeinfo"]).decode().split("\n")
    info = {}
    for line in out:
        if line.strip():
            label, name, h = line.split()
            info[label] = (name, int(h))
    return info


def write_uty(path, name, hash_code, state, data):
    """The reference's on-disk format (include/Unity.cuh:924-971)."""
    raw = np.ascontiguousarray(data)
    with open(path, "wb") as f:
        f.write(name.encode() + b"\n")
        f.write(struct.pack("<Q", hash_code) + b"\n")
        f.write(struct.pack("<iQ", state, len(raw)) + b"\n")
        f.write(raw.tobytes())


def read_uty(path, dtype):
    raw = open(path, "rb").read()
    nl = raw.index(b"\n")
    p = nl + 1 + 8 + 1
    state, count = struct.unpack_from("<iQ", raw, p)
    p += 12 + 1
    return np.frombuffer(raw, dtype, count, p).copy(), state, raw[:nl].decode()


def test_typeid_names_and_hashes_match_reference_checkpoints():
    ref = json.load(open(os.path.join(H.GOLDEN, "uty_headers.json")))
    info = typeinfo()
    for label in ("uchar", "float3", "KeyPoint", "MultiMatch", "Feature"):
        name, h = info[label]
        assert name in ref and ref[name] == h, (label, name, h)
    assert info["Image"][0] == "N6ssrlcv5ImageE"  # .cpimg file names


def test_unity_state_machine_and_checkpoints_cpu(tmp_path):
    info = typeinfo()
    d = str(tmp_path)
    pts = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], np.float32)
    write_uty(os.path.join(d, "0_6float3.uty"), *info["float3"], 1, pts.view(np.dtype([("v", "<f4", (3,))])).reshape(-1))
    v = np.load(os.path.join(H.GOLDEN, "Pipeline2View.npz"))
    open(os.path.join(d, "0_N6ssrlcv5ImageE.cpimg"), "wb").write(v["cpimg_raw"][0].tobytes())
    out = subprocess.check_output([build_binary(), "cpu", d]).decode()
    assert "cpu ok" in out
    # the file written by Unity<KeyPoint>::checkpoint has the reference layout
    kp, state, name = read_uty(os.path.join(d, "7_N6ssrlcv8KeyPointE.uty"), H.KEYPOINT)
    assert name == "N6ssrlcv8KeyPointE" and state == 1 and len(kp) == 6
    assert list(kp["parentId"]) == [3, 3, 4, 4, 5, 5]


def _prepare_pipeline_dir(d, views):
    info = typeinfo()
    pix = H.load_everest_pixels()
    v = np.load(os.path.join(H.GOLDEN, "Pipeline%dView.npz" % views))
    for i in range(views):
        write_uty(os.path.join(d, "pixels_%d.uty" % i), *info["uchar"], 1, pix[i].reshape(-1))
        open(os.path.join(d, "%d_N6ssrlcv5ImageE.cpimg" % i), "wb").write(v["cpimg_raw"][i].tobytes())
    seed, _ = H.load_seed_features()
    write_uty(os.path.join(d, "-1_%s.uty" % info["Feature"][0]), *info["Feature"], 2, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("views", [2, 3])
def test_reference_stage_flow_through_class_api(tmp_path, views):
    d = str(tmp_path)
    _prepare_pipeline_dir(d, views)
    out = subprocess.check_output([build_binary(), "pipeline%d" % views, d]).decode()
    assert "pipeline ok" in out, out
    gv = H.load_view("Pipeline%dView" % views)
    kp, _, _ = read_uty(os.path.join(d, "100_N6ssrlcv8KeyPointE.uty"), H.KEYPOINT)
    mm, _, _ = read_uty(os.path.join(d, "100_N6ssrlcv10MultiMatchE.uty"), H.MULTIMATCH)
    pts, state, _ = read_uty(os.path.join(d, "100_6float3.uty"), np.dtype(("<f4", (3,))))
    assert state == 1  # triangulators return the cloud on the cpu
    ref_kp, ref_mm = gv["kp0"], gv["mm0"]
    # descriptor bytes can differ from the CUDA build by 1 LSB (libm), flipping a handful of borderline ratio tests
    assert abs(len(mm) - len(ref_mm)) <= 0.01 * len(ref_mm), (len(mm), len(ref_mm))
    def groups(kparr, mmarr):
        out = set()
        for n, idx in zip(mmarr["numKeyPoints"], mmarr["index"]):
            out.add(tuple((int(k["parentId"]), float(k["loc"][0]), float(k["loc"][1])) for k in kparr[idx: idx + n]))
        return out
    got, ref = groups(kp, mm), groups(ref_kp, ref_mm)
    assert len(got & ref) >= 0.985 * len(ref), (len(got & ref), len(ref))
    assert len(pts) == len(mm) and np.isfinite(pts).all()
    if len(mm) == len(ref_mm) and np.array_equal(kp["loc"], ref_kp["loc"]):
        diff = pts - gv["points0"]
        rms = float(np.sqrt((diff.astype(np.float64) ** 2).sum(1).mean()))
        assert rms <= (1e-4 if views == 2 else 2.5e-3)
    if views == 2:
        adj, _, _ = read_uty(os.path.join(d, "101_6float3.uty"), np.dtype(("<f4", (3,))))
        # BundleAdjustTwoView is an identity on the cloud upstream (2_6float3.uty == 1_6float3.uty)
        assert adj.shape == pts.shape
        assert np.abs(adj - pts).max() <= 5e-4
