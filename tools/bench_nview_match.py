"""Developer timing of the match stage of the N-view flow (config[3]: six band-culled pairs of four 4096^2 views).
usage: [SSRLCV_HIP_LIB=variant.so] python tools/bench_nview_match.py [--size 4096] [--views 4] [--iters 5]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import helpers as H  # noqa: E402

H.limit_openmp()
from ssrlcv_amd import capi, pipeline  # noqa: E402
import scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--views", type=int, default=4)
ap.add_argument("--iters", type=int, default=5)
args = ap.parse_args()
imgs, cams, _, _ = scene.pinhole_views(args.views, args.size)
seed, _ = H.load_seed_features()
ws = pipeline.Workspace()
res = pipeline.reconstruct(imgs, cams, seed_features=seed, mode=1, ws=ws)
feats = res["features"]
times = []
for _ in range(args.iters):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pairs = pipeline.match_pairs(feats, cams, seed, 25.0, 5.0, mode=1, ws=ws)
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) * 1e3)
times.sort()
lib = capi.LIB
if hasattr(lib, "ssrlcv_dbg_match_stats"):  # a -DSSRLCV_MATCH_STATS build (make -C ssrlcv_amd/csrc instrumented; SSRLCV_HIP_LIB=ssrlcv_amd/libssrlcv_hip_instrumented.so)
    import ctypes
    out = (ctypes.c_ulonglong * 12)()
    lib.ssrlcv_dbg_match_stats(out)  # clear
    pipeline.match_pairs(feats, cams, seed, 25.0, 5.0, mode=1, ws=ws)
    lib.ssrlcv_dbg_match_stats(out)
    names = ["super tests", "group tests", "tile tests", "chains", "slow chains", "candidate rows", "lane candidates (acc)",
             "lane candidates (exact v)", "passing"]
    print("walk counters of one match stage: " + ", ".join("%s %.3f M" % (n, out[i] / 1e6) for i, n in enumerate(names)))
    print("most chains of one wave %d; waves with > 500 chains %d, > 2000 chains %d" % (out[9], out[10], out[11]))
print("match stage (%d pairs, features %s): min %.2f median %.2f ms; %d matches" % (
    len(pairs), [f.numel() // 152 for f in feats], times[0], times[len(times) // 2], sum(p.numel() // 16 for p in pairs.values())))
