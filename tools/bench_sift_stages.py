
"""Developer timing of the SIFT stages on one synthetic image (not part of the driver contract).
usage: [SSRLCV_HIP_LIB=variant.so] python tools/bench_sift_stages.py [--size 4096]
Prints ms of ssrlcv_hip_sift_build_dog and of ssrlcv_hip_sift_describe stopped after each stage (differences = stage
cost: 1 extrema, 2 refine+rescan, 3 noise, 4 edges, 5 window check, 6 orientations, 7 descriptors)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ssrlcv_amd import capi  # noqa: E402
import bench  # noqa: E402


def timeit(fn, iters=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--stages", default="1,2,3,4,5,6,7")
    ap.add_argument("--scene", action="store_true", help="a tools/scene.py view (realistic feature density) instead of the dense noise image")
    args = ap.parse_args()
    S = args.size
    if args.scene:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import scene
        img = scene.pinhole_views(1, S)[0][0]
    else:
        img = bench.synth_images(1, S, S, seed=0, device="cuda")[0]
    plan = capi.SiftPlan(S, S)
    print("build_dog  %.3f ms" % timeit(lambda: plan.build_dog(img)))
    prev = 0.0
    for st in [int(s) for s in args.stages.split(",")]:
        plan.set_stop_stage(st)
        ms = timeit(plan.describe)
        print("describe stop=%d  %.3f ms  (+%.3f)  n=%d" % (st, ms, ms - prev, plan.count()))
        prev = ms


if __name__ == "__main__":
    main()
