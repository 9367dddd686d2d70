"""N-view reconstruction flow (BASELINE.json configs[1..4]) over the HIP C ABI, single- or multi-GPU.

Mirrors src/Pipeline.cu doFeatureGeneration -> doFeatureMatching -> doTriangulation (+ the bundle-adjustment error
sweep of doBundleAdjust) with the sharding of ssrlcv_amd/dist.py:

  stage A   SIFT per image on its owner rank                         exchange 1: all-gather of the feature arrays
  stage B   pair matching on the pair's owner rank                   exchange 2: all-gather of the uint2_pair arrays
  merge     generateMatchesExhaustive's host merge, replicated (deterministic)
  stage C   bundle-range triangulation                                exchange 3: all-gather of the cloud
  BA sweep  the K finite-difference evaluations of f(cameras) over this rank's bundle range of the first image pair
            (calculateImageGradient / Hessian, src/PointCloudFactory.cu:1059-1504)   all-reduce(sum) of the K sums

With world size 1 (no process group) every exchange is the identity.  Python here is plumbing (device buffers,
torch.distributed); all compute is libssrlcv_hip.so.  A `Workspace` keeps the per-rank plans, matcher workspaces and
seed descriptors between calls: creating and freeing them per pair / per image was a third of a step.
"""
import time

import numpy as np
import torch
import torch.distributed as tdist

from . import capi
from . import dist as sd

KEYPOINT = np.dtype([("parentId", "<i4"), ("pad", "<i4"), ("loc", "<f4", (2,))])
MULTIMATCH = np.dtype([("numKeyPoints", "<u4"), ("index", "<i4")])
FEATURE_BYTES = 152


def _world():
    if tdist.is_available() and tdist.is_initialized():
        return tdist.get_world_size(), tdist.get_rank()
    return 1, 0


class Workspace:
    """Per-rank cache: SIFT plans by image size, one grow-only matcher workspace / output buffer, the seed features on
    the device.  One flow at a time per Workspace."""

    def __init__(self):
        self.plans = {}
        self.match_ws = None
        self.match_out = None
        self.seed = None
        self.seed_key = None
        self.times = {}

    def plan(self, w, h, slot=0):
        key = (w, h, slot)
        if key not in self.plans:
            self.plans[key] = capi.SiftPlan(w, h)
        return self.plans[key]

    def matcher(self, shapes, out_bytes):
        need = max(capi.LIB.ssrlcv_hip_match_workspace_bytes(capi.c_u32(nq), capi.c_u32(nt)) for nq, nt in shapes)
        if self.match_ws is None or self.match_ws.numel() < need:
            self.match_ws = capi.dev_bytes(need)
        if self.match_out is None or self.match_out.numel() < out_bytes:
            self.match_out = capi.dev_bytes(out_bytes)
        return self.match_ws, self.match_out

    def seed_features(self, seed_np):
        if seed_np is None:
            return None
        key = (id(seed_np), len(seed_np))
        if self.seed_key != key:
            self.seed, self.seed_key = capi.to_dev(seed_np), key
        return self.seed

    def tick(self, name, t0):
        self.times[name] = self.times.get(name, 0.0) + (time.perf_counter() - t0)


def extract_features(pixel_tensors, ws=None):
    """SIFT on the images this rank owns.  pixel_tensors: {image index: u8 CUDA tensor (H, W)} -> {index: feature bytes}.
    The extracts are queued back to back (one plan per owned image) and synchronised once."""
    ws = ws or Workspace()
    plans = {}
    for k, (v, pix) in enumerate(pixel_tensors.items()):
        h, w = pix.shape
        plans[v] = ws.plan(w, h, k)
        plans[v].extract(pix)
    out = {}
    for v, plan in plans.items():
        n = plan.count()   # synchronises; raises if a key-point list overflowed
        out[v] = plan.features[: n * FEATURE_BYTES].clone()
    return out


def exchange_features(local, num_images):
    world, _ = _world()
    if world == 1:
        return [local[v] for v in range(num_images)]
    return sd.exchange_keyed(local, num_images, sd.image_owner)


def match_pairs(features, cameras, seed_features=None, epsilon=25.0, delta=5.0, rel=0.6, absolute=200.0 * 200.0, mode=1,
                ws=None):
    """Exhaustive matching (generateMatchesExhaustive; GEO_ORBIT double-constrained for mode 1, brute force for mode 0)
    of the pairs this rank owns.  features: list of uint8 CUDA tensors (all images).  Returns {pair index: validated
    uint2_pair bytes}."""
    ws = ws or Workspace()
    world, rank = _world()
    num_images = len(features)
    pairs = sd.pair_list(num_images)
    seed_d = ws.seed_features(seed_features)
    out = {}
    seed_cache = {}
    for p, (qi, ti) in enumerate(pairs):
        if sd.pair_owner(p, world) != rank:
            continue
        nq, nt = features[qi].numel() // FEATURE_BYTES, features[ti].numel() // FEATURE_BYTES
        shapes = [(nq, nt)] + ([(nq, len(seed_features))] if seed_features is not None else [])
        mws, mout = ws.matcher(shapes, nq * 16)
        sdist = None
        if seed_d is not None:
            if qi not in seed_cache:  # recomputed per query image upstream (src/MatchFactory.cu:925)
                seed_cache[qi] = capi.seed_distances(features[qi], nq, seed_d, len(seed_features), workspace=mws)
            sdist = seed_cache[qi]
        proj = capi.projection_matrix(cameras[ti:ti + 1]) if mode == 1 else None
        params = capi.make_match_params(mode, qi, ti, epsilon, delta, rel, absolute,
                                        cameras[qi:qi + 1] if mode == 1 else None, proj)
        res = capi.match(features[qi], nq, features[ti], nt, params, capi.OUT_UINT2_PAIR, seed_d=sdist, workspace=mws,
                         out=mout)
        n = capi.compact_matches(capi.OUT_UINT2_PAIR, res, nq, mws)
        out[p] = res[: n * 16].clone()
    return out


def exchange_pairs(local, num_pairs):
    world, _ = _world()
    if world == 1:
        return [local[p] for p in range(num_pairs)]
    return sd.exchange_keyed(local, num_pairs, sd.pair_owner)


def build_match_set(features, pair_tensors):
    """Replicated host merge -> (MultiMatch numpy, KeyPoint numpy)."""
    num_features = [f.numel() // FEATURE_BYTES for f in features]
    mm, mem = sd.merge_matches(num_features, pair_tensors)
    kp = np.zeros(len(mem), KEYPOINT)
    kp["parentId"] = mem[:, 0]
    if len(mem):
        # key-point locations (byte 8 of each 152-byte feature) gathered on the device: one small D2H copy instead of
        # every image's full location table
        mem_d = torch.from_numpy(mem.astype(np.int64)).to(features[0].device)
        loc = torch.empty(len(mem), 2, dtype=torch.float32, device=features[0].device)
        for v, f in enumerate(features):
            sel = (mem_d[:, 0] == v).nonzero().squeeze(1)
            if sel.numel():
                table = f.view(-1, FEATURE_BYTES)[:, 8:16].contiguous().view(torch.float32).view(-1, 2)
                loc[sel] = table[mem_d[sel, 1]]
        kp["loc"] = loc.cpu().numpy()
    return mm, kp


def triangulate(mm, kp, cameras, nview, pushbroom=None):
    """Bundle-range partitioned triangulation; every rank ends with the full cloud.  `pushbroom`: PushbroomCamera array
    (config[4]): bundles then come from generatePushbroomBundle (src/PointCloudFactory.cu:875-903)."""
    world, rank = _world()
    lo, hi = sd.bundle_range(len(mm), world, rank)
    sub = mm[lo:hi].copy()
    n = len(sub)
    pts = torch.zeros(0, dtype=torch.float32, device="cuda")
    if n:
        if pushbroom is not None:
            b_d, l_d = capi.generate_pushbroom_bundles(capi.to_dev(sub), capi.to_dev(kp), n, capi.to_dev(pushbroom),
                                                       len(pushbroom), len(kp))
        else:
            b_d, l_d = capi.generate_bundles(capi.to_dev(sub), capi.to_dev(kp), n, capi.to_dev(cameras), len(cameras), len(kp))
        pts, _, _ = capi.triangulate(l_d, b_d, n, nview=nview)
    if world == 1:
        return pts.view(-1, 3)
    parts = sd.all_gather_bytes(pts.view(torch.uint8).reshape(-1).contiguous())
    return torch.cat([p.view(torch.float32) for p in parts]).view(-1, 3)


def ba_parameter_sets(cameras2, h_lin=1e-5, h_step=(1e-4, 1e-4, 1e-4, 1e-5, 1e-5, 1e-5)):
    """The K = 612 camera-parameter vectors one BundleAdjustTwoView iteration evaluates: 24 for the central-difference
    gradient (h = 1e-5, src/PointCloudFactory.cu:1061-1062) and 588 for the Hessian (h = {1e-4 x3, 1e-5 x3} per camera,
    :1261): 12 x 5 points of the diagonal stencil (:1375-1376) and 132 ordered parameter pairs x 4 points of the cross
    stencil (:1444-1445).  Layout per set: camera-major {pos.xyz, rot.xyz}.  (The exact evaluation order and the float
    drift of upstream's in-place perturbation live in the C++ mirror, host/PointCloudFactory.hpp; this builds the same
    612-point workload for the sharded sweep.)"""
    base = np.concatenate([np.concatenate([c["cam_pos"], c["cam_rot"]]) for c in cameras2]).astype(np.float32)
    hs = np.array(list(h_step) * 2, np.float32)
    sets = []
    for i in range(12):
        for s in (+1, -1):
            p = base.copy()
            p[i] += s * np.float32(h_lin)
            sets.append(p)
    for i in range(12):
        for m in (-2, -1, 0, 1, 2):
            p = base.copy()
            p[i] += m * hs[i]
            sets.append(p)
    for i in range(12):
        for j in range(12):
            if i == j:
                continue
            for si, sj in ((1, 1), (1, -1), (-1, 1), (-1, -1)):
                p = base.copy()
                p[i] += si * hs[i]
                p[j] += sj * hs[j]
                sets.append(p)
    assert len(sets) == 612
    return np.stack(sets)


def ba_error_sweep(mm, kp, cameras, pair=(0, 1), params=None):
    """f(params_k) = sum of squared skew-line gaps of the 2-view bundles of `pair`, for all K parameter sets: this rank's
    bundle range in one launch (ssrlcv_hip_ba_sweep2), then all-reduce(sum) over the ranks.  Returns a K-vector on the
    device (identical on every rank up to the float all-reduce)."""
    world, rank = _world()
    two = np.nonzero(mm["numKeyPoints"] == 2)[0]
    if len(two):
        first = kp["parentId"][mm["index"][two]]
        second = kp["parentId"][mm["index"][two] + 1]
        two = two[(first == pair[0]) & (second == pair[1])]
    cams2 = cameras[[pair[0], pair[1]]].copy()
    if params is None:
        params = ba_parameter_sets(cams2)
    K = len(params)
    lo, hi = sd.bundle_range(len(two), world, rank)
    sums = torch.zeros(K, dtype=torch.float32, device="cuda")
    if hi > lo:
        idx = mm["index"][two[lo:hi]]
        sub_kp = np.zeros(2 * (hi - lo), KEYPOINT)
        sub_kp["parentId"][0::2], sub_kp["parentId"][1::2] = 0, 1
        sub_kp["loc"][0::2], sub_kp["loc"][1::2] = kp["loc"][idx], kp["loc"][idx + 1]
        sub_mm = np.zeros(hi - lo, MULTIMATCH)
        sub_mm["numKeyPoints"], sub_mm["index"] = 2, 2 * np.arange(hi - lo)
        sums = capi.ba_sweep2(capi.to_dev(sub_mm), capi.to_dev(sub_kp), hi - lo, capi.to_dev(cams2), 2,
                              torch.from_numpy(np.ascontiguousarray(params, np.float32)).cuda(), K)
    if world > 1:
        sd.all_reduce_sum(sums)
    return sums, len(two)


def reconstruct(pixel_tensors_all, cameras, seed_features=None, epsilon=25.0, delta=5.0, mode=1, ws=None, pushbroom=None,
                ba=False):
    """Full flow.  pixel_tensors_all: list of u8 CUDA tensors (only the owner rank's entries are used).  `ws`: a
    Workspace kept by the caller between calls; its `times` dict accumulates the wall time of every stage."""
    ws = ws or Workspace()
    world, rank = _world()
    num_images = len(pixel_tensors_all)
    mine = {v: pixel_tensors_all[v] for v in range(num_images) if sd.image_owner(v, world) == rank}
    t = time.perf_counter()
    local = extract_features(mine, ws)
    ws.tick("sift", t)
    t = time.perf_counter()
    feats = exchange_features(local, num_images)
    ws.tick("exchange_features", t)
    t = time.perf_counter()
    pair_local = match_pairs(feats, cameras, seed_features, epsilon, delta, mode=mode, ws=ws)
    torch.cuda.synchronize()
    ws.tick("match", t)
    t = time.perf_counter()
    pair_all = exchange_pairs(pair_local, len(sd.pair_list(num_images)))
    ws.tick("exchange_pairs", t)
    t = time.perf_counter()
    mm, kp = build_match_set(feats, pair_all)
    ws.tick("merge", t)
    t = time.perf_counter()
    cloud = triangulate(mm, kp, cameras, nview=num_images > 2, pushbroom=pushbroom)
    torch.cuda.synchronize()
    ws.tick("triangulate", t)
    out = {"features": feats, "pairs": pair_all, "matches": mm, "keypoints": kp, "points": cloud}
    if ba and pushbroom is None:
        t = time.perf_counter()
        out["ba_sums"], out["ba_bundles"] = ba_error_sweep(mm, kp, cameras)
        torch.cuda.synchronize()
        ws.tick("ba_sweep", t)
    return out
