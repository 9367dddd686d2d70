"""Developer timing of the device merge stage of the N-view flow (config[3]) with its sub-steps separated.
usage: python tools/bench_merge_device.py [--size 4096] [--views 4] [--iters 4]"""
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
from ssrlcv_amd import capi, pipeline  # noqa: E402
import scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--iters", type=int, default=4)
    args = ap.parse_args()
    imgs, cams, _, _ = scene.pinhole_views(args.views, args.size)
    seed, _ = H.load_seed_features()
    ws = pipeline.Workspace()
    res = pipeline.reconstruct(imgs, cams, seed_features=seed, mode=1, ws=ws, ba=False)
    feats, pairs = res["features"], res["pairs"]
    nf = [f.numel() // 152 for f in feats]
    counts = [p.numel() // 16 for p in pairs]
    print("features", nf, "pairs", counts, "multi-matches", len(res["matches"]))

    def lap(name, t0):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print("  %-40s %7.3f ms" % (name, (t1 - t0) * 1e3))
        return t1
    mws = None
    for it in range(args.iters):
        print("iteration", it)
        t = time.perf_counter()
        live = [p.reshape(-1) for p in pairs if p.numel()]
        cat = torch.cat(live)
        t = lap("torch.cat(pairs)", t)
        out = capi.merge_matches_device(nf, counts, cat, mws)
        mm_d, mem_d, n_mm, n_mem, rounds, mws = out[:6]
        t = lap("ssrlcv_hip_merge_matches (%s rounds)" % rounds, t)
        kp_d = capi.keypoints_from_members(mem_d, n_mem, feats)
        t = lap("keypoints_from_members", t)
        mm = pipeline._host_records(mm_d, n_mm, pipeline.MULTIMATCH, "mm")
        t = lap("MultiMatch D2H + host copy (%d)" % n_mm, t)
        kp = pipeline._host_records(kp_d, n_mem, pipeline.KEYPOINT, "kp")
        t = lap("KeyPoint D2H + host copy (%d)" % n_mem, t)
        t0 = time.perf_counter()
        pipeline.build_match_set(feats, pairs, {})
        lap("build_match_set total", t0)


if __name__ == "__main__":
    main()
