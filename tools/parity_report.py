"""Prints how closely the HIP SIFT path reproduces the CPU oracle and the reference's golden match sets:
feature counts, bit-equality of loc / sigma / theta, descriptor byte differences, and the match-set overlap with the
13 534 (2-view) / 21 177 (3-view) golden sets.  Run on a GPU box from the repository root:
    python3 tools/parity_report.py [--size 1024]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from ssrlcv_amd import capi  # noqa: E402


def compare(name, gf, of):
    print("%-22s features HIP %d / oracle %d" % (name, len(gf), len(of)))
    if len(gf) != len(of):
        return False
    ok = True
    for f in ("loc", "sigma", "theta"):
        eq = gf[f].view(np.uint32) == of[f].view(np.uint32)
        print("    %-6s bit-equal %d / %d" % (f, int(eq.sum()), eq.size))
        ok &= bool(eq.all())
    d = gf["values"].astype(np.int32) - of["values"].astype(np.int32)
    nd = int((d != 0).any(1).sum())
    print("    descriptors differing %d / %d, bytes differing %d, max |diff| %d" % (nd, len(gf), int((d != 0).sum()),
                                                                                 int(np.abs(d).max()) if d.size else 0))
    return ok and nd == 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    args = ap.parse_args()
    lib = H.oracle()
    all_ok = True
    imgs = {"synthetic %d" % args.size: H.synthetic_image(args.size, args.size, seed=3),
            "synthetic 384x256": H.synthetic_image(384, 256, seed=1)}
    pix = H.load_everest_pixels()
    for i, p in enumerate(pix):
        imgs["everest %d" % i] = p
    feats = {}
    for name, img in imgs.items():
        h, w = img.shape
        plan = capi.SiftPlan(w, h)
        plan.extract(capi.to_dev(img))
        gf = plan.features_host(H.FEATURE)
        t0 = time.time()
        of = H.oracle_sift(lib, img)
        all_ok &= compare(name, gf, of)
        feats[name] = (plan, gf, of)
        print("    (oracle %.1f s)" % (time.time() - t0))
    # golden match sets through the HIP path
    seed, _ = H.load_seed_features()
    v = H.load_view("Pipeline2View")
    cams = v["cameras"]
    p0, p1 = feats["everest 0"][0], feats["everest 1"][0]
    n0, n1 = p0.count(), p1.count()
    sd_d = capi.seed_distances(p0.features, n0, capi.to_dev(seed), len(seed))
    params = capi.make_match_params(1, 0, 1, 25.0, 5.0, 0.6, 200.0 * 200.0, cams[0:1], capi.projection_matrix(cams[1:2]))
    out_d = capi.match(p0.features, n0, p1.features, n1, params, capi.OUT_DMATCH, seed_d=sd_d)
    n = capi.compact_matches(capi.OUT_DMATCH, out_d, n0, capi.match_workspace(n0, n1))
    dm = capi.to_host(out_d, H.DMATCH, n)
    kp = v["kp0"]
    ref_pairs = {(tuple(a), tuple(b)) for a, b in zip(kp["loc"][0::2].tolist(), kp["loc"][1::2].tolist())}
    got_pairs = {(tuple(a), tuple(b)) for a, b in zip(dm["kp0_loc"].tolist(), dm["kp1_loc"].tolist())}
    same_order = n == 13534 and np.array_equal(dm["kp0_loc"], kp["loc"][0::2]) and np.array_equal(dm["kp1_loc"], kp["loc"][1::2])
    print("2-view golden: HIP %d matches, %d / %d golden pairs reproduced, identical list: %s" %
          (n, len(ref_pairs & got_pairs), len(ref_pairs), same_order))
    all_ok &= bool(same_order)
    from ssrlcv_amd import pipeline
    v3 = H.load_view("Pipeline3View")
    res = pipeline.reconstruct([capi.to_dev(p).view(1024, 1024) for p in pix], v3["cameras"], seed_features=seed)
    mm, kp3 = res["matches"], res["keypoints"]
    same3 = len(mm) == 21177 and np.array_equal(mm["numKeyPoints"], v3["mm0"]["numKeyPoints"]) and \
        np.array_equal(kp3["loc"], v3["kp0"]["loc"]) and np.array_equal(kp3["parentId"], v3["kp0"]["parentId"])
    print("3-view golden: HIP %d multi-matches (golden 21177), identical structure + key points: %s" % (len(mm), same3))
    all_ok &= bool(same3)
    print("PARITY", "EXACT" if all_ok else "NOT EXACT")


if __name__ == "__main__":
    main()
