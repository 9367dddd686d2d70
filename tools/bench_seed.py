"""Developer timing of ssrlcv_hip_seed_distances_u8x128 (getSeedDistances: every feature of an image against a few
thousand seed features).  usage: python tools/bench_seed.py [--nq 629559] [--ns 1116,4096,16384]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from ssrlcv_amd import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nq", type=int, default=629559)
ap.add_argument("--ns", default="1116,4096,16384")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--scene", action="store_true", help="real features of a rendered 4096^2 view against the fixture's seed set")
args = ap.parse_args()
if args.scene:
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scene
    from ssrlcv_amd import pipeline
    imgs, cams, _, _ = scene.pinhole_views(3, 4096)
    plan = capi.SiftPlan(4096, 4096)
    plan.extract(imgs[2])
    nq = plan.count()
    feats = plan.features
    seed, _ = H.load_seed_features()
    sd = capi.to_dev(seed)
    ws = capi.match_workspace(nq, len(seed))
    for rep in range(2):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            out = capi.seed_distances(feats, nq, sd, len(seed), ws)
        e1.record()
        torch.cuda.synchronize()
    print("scene: nq %d ns %d: %.3f ms per call" % (nq, len(seed), e0.elapsed_time(e1) / args.iters))
    sys.exit(0)
rng = np.random.default_rng(5)


def random_features(n):
    f = np.zeros(n, H.FEATURE)
    f["values"] = rng.integers(0, 256, (n, 128), dtype=np.uint8)
    f["loc"] = rng.uniform(0, 4096, (n, 2)).astype(np.float32)
    return f


q = random_features(args.nq)
qd = capi.to_dev(q)
for ns in [int(x) for x in args.ns.split(",")]:
    s = random_features(ns)
    sd = capi.to_dev(s)
    ws = capi.match_workspace(args.nq, ns)
    out = capi.seed_distances(qd, args.nq, sd, ns, ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        out = capi.seed_distances(qd, args.nq, sd, ns, ws)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    print("nq %d ns %d: %.3f ms per call (%.2f POP/s)" % (args.nq, ns, ms, 2.0 * 128 * args.nq * ns / ms / 1e12))
