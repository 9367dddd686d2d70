"""Developer timing of ssrlcv_hip_sift_build_dog alone (the pyramid stage's time does not depend on the image content).
usage: [SSRLCV_HIP_LIB=variant.so] python3 tools/bench_pyramid.py [--size 4096] [--iters 10]
Under `rocprofv3 --kernel-trace` the rocpd database can be turned into a timeline of the last call with
tools/timeline.py <results.db> pyramid."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ssrlcv_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    S = args.size
    g = torch.Generator(device="cuda").manual_seed(1)
    img = torch.randint(0, 256, (S, S), dtype=torch.uint8, device="cuda", generator=g)
    plan = capi.SiftPlan(S, S)
    for _ in range(2):
        plan.build_dog(img)
    torch.cuda.synchronize()
    times = []
    for _ in range(args.iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        plan.build_dog(img)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    times.sort()
    print("build_dog %dx%d: min %.3f  median %.3f  max %.3f ms" % (S, S, times[0], times[len(times) // 2], times[-1]), flush=True)


if __name__ == "__main__":
    main()
