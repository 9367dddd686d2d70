"""Developer timing of the N-view flow's host-side stages (config[3]: 4-view 4096^2) with the sub-steps of the merge and
the triangulation separated.  usage: [SSRLCV_MERGE_TIMING=1] python3 tools/bench_nview.py [--size 4096] [--views 4]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import helpers as H  # noqa: E402

H.limit_openmp()
from ssrlcv_amd import capi, pipeline, dist as sd  # noqa: E402
import scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--iters", type=int, default=3)
    args = ap.parse_args()
    imgs, cams, _, _ = scene.pinhole_views(args.views, args.size)
    seed, _ = H.load_seed_features()
    ws = pipeline.Workspace()
    res = pipeline.reconstruct(imgs, cams, seed_features=seed, mode=1, ws=ws, ba=True)
    feats, pairs = res["features"], res["pairs"]
    print("features", [f.numel() // 152 for f in feats], "pairs", [p.numel() // 16 for p in pairs], "multi-matches", len(res["matches"]))

    def lap(name, t0):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print("  %-34s %7.2f ms" % (name, (t1 - t0) * 1e3))
        return t1
    for it in range(args.iters):
        print("iteration", it)
        t = time.perf_counter()
        nf = [f.numel() // 152 for f in feats]
        live = [p.reshape(-1) for p in pairs if p.numel()]
        cat = torch.cat(live)
        t = lap("torch.cat(pairs)", t)
        host = cat.cpu().numpy()
        t = lap("pairs D2H", t)
        mm, mem = sd.merge_matches(nf, pairs)
        t = lap("merge_matches (cat + D2H + C merge)", t)
        mem_d = capi.to_dev(np.ascontiguousarray(mem, np.uint32))
        t = lap("members H2D", t)
        kp_d = capi.keypoints_from_members(mem_d, len(mem), feats)
        t = lap("keypoints_from_members", t)
        kp = capi.to_host(kp_d, pipeline.KEYPOINT, len(mem))
        t = lap("keypoints D2H", t)
        sub_d = capi.to_dev(mm)
        t = lap("multi-matches H2D", t)
        b_d, l_d = capi.generate_bundles(sub_d, kp_d, len(mm), capi.to_dev(cams), len(cams), len(kp))
        t = lap("generate_bundles", t)
        pts, _, _ = capi.triangulate(l_d, b_d, len(mm), nview=True)
        t = lap("triangulateN", t)
    ws.times.clear()
    for _ in range(3):
        pipeline.reconstruct(imgs, cams, seed_features=seed, mode=1, ws=ws, ba=True)
    print({k: round(v / 3 * 1e3, 2) for k, v in ws.times.items()})


if __name__ == "__main__":
    main()
