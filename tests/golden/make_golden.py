#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the reference's own test checkpoints.

Reads the binary `.uty` / `.cpimg` fixtures under
/root/reference/test/checkpoints (format: include/Unity.cuh:924-971 and
src/Image.cu:274-303) and stores them as compressed numpy archives.  The
archives are DATA (inputs + expected outputs the reference's gtest compares
against, test/Pipeline.cu:104-436); no reference source text is stored.

Run in the build container only (the GPU box has no /root/reference):
    python tests/golden/make_golden.py
"""
import os
import sys
import numpy as np

REF = "/root/reference/test/checkpoints"
OUT = os.path.dirname(os.path.abspath(__file__))

KEYPOINT = np.dtype([("parentId", "<i4"), ("pad", "<i4"), ("loc", "<f4", (2,))])
assert KEYPOINT.itemsize == 16
MULTIMATCH = np.dtype([("numKeyPoints", "<u4"), ("index", "<i4")])
FLOAT3 = np.dtype(("<f4", (3,)))
FEATURE = np.dtype([("parent", "<i4"), ("pad", "<i4"), ("loc", "<f4", (2,)),
                    ("sigma", "<f4"), ("theta", "<f4"), ("values", "u1", (128,))])
assert FEATURE.itemsize == 152

# Image::Camera (include/Image.cuh:40-57), 80 bytes, offsets measured with the
# HIP toolchain.
CAMERA = np.dtype({
    "names": ["cam_pos", "cam_rot", "fov", "foc", "dpix", "timeStamp",
              "ecef_offset", "no_rot", "size"],
    "formats": [("<f4", (3,)), ("<f4", (3,)), ("<f4", (2,)), "<f4", ("<f4", (2,)),
                "<i8", ("<f4", (3,)), "u1", ("<u4", (2,))],
    "offsets": [0, 12, 24, 32, 40, 48, 56, 68, 72],
    "itemsize": 80})


def read_uty(path, dtype):
    with open(path, "rb") as f:
        raw = f.read()
    nl = raw.index(b"\n")
    name = raw[:nl].decode()
    p = nl + 1
    p += 8 + 1            # hash_code + eol
    state = int(np.frombuffer(raw, "<i4", 1, p)[0]); p += 4
    count = int(np.frombuffer(raw, "<u8", 1, p)[0]); p += 8
    p += 1                # eol
    arr = np.frombuffer(raw, dtype, count, p).copy()
    assert p + count * np.dtype(dtype).itemsize == len(raw), (path, name)
    return name, state, arr


def read_uty_header(path):
    raw = open(path, "rb").read(256)
    nl = raw.index(b"\n")
    return raw[:nl].decode(), int(np.frombuffer(raw, "<u8", 1, nl + 1)[0])


def read_cpimg(path):
    raw = open(path, "rb").read()
    assert len(raw) == 240
    img_id = int(np.frombuffer(raw, "<i4", 1, 32)[0])
    size = np.frombuffer(raw, "<u4", 2, 40).copy()
    color = int(np.frombuffer(raw, "<u4", 1, 48)[0])
    cam = np.frombuffer(raw, CAMERA, 1, 56).copy()
    return img_id, size, color, cam


def main():
    for view, nimg in (("Pipeline2View", 2), ("Pipeline3View", 3)):
        d = os.path.join(REF, view)
        out = {}
        cams = []
        for i in range(nimg):
            img_id, size, color, cam = read_cpimg(os.path.join(d, "%d_N6ssrlcv5ImageE.cpimg" % i))
            assert img_id == i and color == 1
            cams.append(cam)
            out["image_size_%d" % i] = size
        cams = np.concatenate(cams)
        # plain per-field arrays: numpy repacks offset-bearing structured dtypes on save
        for name in ("cam_pos", "cam_rot", "fov", "foc", "dpix", "ecef_offset", "size"):
            out["cam_" + name.replace("cam_", "")] = np.ascontiguousarray(cams[name])
        for stage in (0, 1):
            _, _, kp = read_uty(os.path.join(d, "%d_N6ssrlcv8KeyPointE.uty" % stage), KEYPOINT)
            _, _, mm = read_uty(os.path.join(d, "%d_N6ssrlcv10MultiMatchE.uty" % stage), MULTIMATCH)
            _, _, pts = read_uty(os.path.join(d, "%d_6float3.uty" % stage), FLOAT3)
            out["kp%d_parent" % stage] = kp["parentId"].copy()
            out["kp%d_loc" % stage] = kp["loc"].copy()
            out["mm%d_num" % stage] = mm["numKeyPoints"].copy()
            out["mm%d_index" % stage] = mm["index"].copy()
            out["points%d" % stage] = pts
        if view == "Pipeline2View":
            _, _, pts = read_uty(os.path.join(d, "2_6float3.uty"), FLOAT3)
            out["points2"] = pts
        # raw 240-byte Image dumps (process-local pointer fields included, ignored by readers) for the .cpimg reader test
        out["cpimg_raw"] = np.stack([np.frombuffer(open(os.path.join(d, "%d_N6ssrlcv5ImageE.cpimg" % i), "rb").read(),
                                                   np.uint8) for i in range(nimg)])
        np.savez_compressed(os.path.join(OUT, view + ".npz"), **out)
        print(view, {k: v.shape for k, v in out.items()})
        # pixels (1024x1024 u8 each)
        pix = {}
        for i in range(nimg):
            _, _, p = read_uty(os.path.join(d, "pixels", "%d_h.uty" % i), "u1")
            pix["pixels_%d" % i] = p.reshape(1024, 1024)
        if view == "Pipeline3View":
            np.savez_compressed(os.path.join(OUT, "everest_pixels.npz"), **pix)
    # typeid name + std::type_info::hash_code of every checkpointed type, as the reference wrote them
    import json
    d2 = os.path.join(REF, "Pipeline2View")
    headers = {}
    for fn in ("0_6float3.uty", "0_N6ssrlcv8KeyPointE.uty", "0_N6ssrlcv10MultiMatchE.uty",
               "-1_N6ssrlcv7FeatureINS_15SIFT_DescriptorEEE.uty", "pixels/0_h.uty"):
        name, h = read_uty_header(os.path.join(d2, fn))
        headers[name] = h
    json.dump(headers, open(os.path.join(OUT, "uty_headers.json"), "w"), indent=1, sort_keys=True)
    print(headers)
    # seed features (identical apart from one byte between the two dirs)
    _, st, f2 = read_uty(os.path.join(REF, "Pipeline2View", "-1_N6ssrlcv7FeatureINS_15SIFT_DescriptorEEE.uty"), FEATURE)
    _, _, f3 = read_uty(os.path.join(REF, "Pipeline3View", "-1_N6ssrlcv7FeatureINS_15SIFT_DescriptorEEE.uty"), FEATURE)
    np.savez_compressed(os.path.join(OUT, "seed_features.npz"),
                        loc=f2["loc"], sigma=f2["sigma"], theta=f2["theta"], values=f2["values"],
                        values_run2=f3["values"], parent=f2["parent"])
    print("seed", f2.shape, "origin state", st, "bytes differing between runs",
          int((f2["values"] != f3["values"]).sum()))


if __name__ == "__main__":
    sys.exit(main())
