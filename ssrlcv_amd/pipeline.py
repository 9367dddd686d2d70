"""N-view reconstruction flow (BASELINE.json configs[1..3]) over the HIP C ABI, single- or multi-GPU.

Mirrors src/Pipeline.cu doFeatureGeneration -> doFeatureMatching -> doTriangulation with the sharding of
ssrlcv_amd/dist.py: SIFT per image on its owner rank, all-gather of features, pair matching on the pair's owner rank,
all-gather of the uint2_pair arrays, replicated host merge, bundle-range triangulation, all-gather of the cloud.
With world size 1 (no process group) every exchange is the identity.  Python here is plumbing (device buffers,
torch.distributed); all compute is libssrlcv_hip.so.
"""
import numpy as np
import torch
import torch.distributed as tdist

from . import capi
from . import dist as sd

KEYPOINT = np.dtype([("parentId", "<i4"), ("pad", "<i4"), ("loc", "<f4", (2,))])
FEATURE_BYTES = 152


def _world():
    if tdist.is_available() and tdist.is_initialized():
        return tdist.get_world_size(), tdist.get_rank()
    return 1, 0


def extract_features(pixel_tensors, plans=None):
    """SIFT on the images this rank owns.  pixel_tensors: {image index: u8 CUDA tensor (H, W)} -> {index: (bytes, n)}"""
    out = {}
    for v, pix in pixel_tensors.items():
        h, w = pix.shape
        plan = plans[v] if plans else capi.SiftPlan(w, h)
        plan.extract(pix)
        n = plan.count()
        out[v] = plan.features[: n * FEATURE_BYTES].clone()
    return out


def exchange_features(local, num_images):
    world, _ = _world()
    if world == 1:
        return [local[v] for v in range(num_images)]
    return sd.exchange_keyed(local, num_images, sd.image_owner)


def match_pairs(features, cameras, seed_features=None, epsilon=25.0, delta=5.0, rel=0.6, absolute=200.0 * 200.0, mode=1):
    """Exhaustive double-constrained matching (generateMatchesExhaustive, GEO_ORBIT path) of the pairs this rank owns.
    features: list of uint8 CUDA tensors (all images).  Returns {pair index: validated uint2_pair bytes}."""
    world, rank = _world()
    num_images = len(features)
    pairs = sd.pair_list(num_images)
    seed_d = capi.to_dev(seed_features) if seed_features is not None else None
    out = {}
    seed_cache = {}
    for p, (qi, ti) in enumerate(pairs):
        if sd.pair_owner(p, world) != rank:
            continue
        nq, nt = features[qi].numel() // FEATURE_BYTES, features[ti].numel() // FEATURE_BYTES
        sdist = None
        if seed_d is not None:
            if qi not in seed_cache:  # recomputed per query image upstream (src/MatchFactory.cu:925)
                seed_cache[qi] = capi.seed_distances(features[qi], nq, seed_d, len(seed_features))
            sdist = seed_cache[qi]
        params = capi.make_match_params(mode, qi, ti, epsilon, delta, rel, absolute, cameras[qi:qi + 1],
                                        capi.projection_matrix(cameras[ti:ti + 1]))
        ws = capi.match_workspace(nq, nt)
        res = capi.match(features[qi], nq, features[ti], nt, params, capi.OUT_UINT2_PAIR, seed_d=sdist, workspace=ws)
        n = capi.compact_matches(capi.OUT_UINT2_PAIR, res, nq, ws)
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
    locs = []
    for f in features:  # loc sits at byte 8 of each 152-byte feature
        locs.append(f.view(-1, FEATURE_BYTES)[:, 8:16].contiguous().view(torch.float32).cpu().numpy())
    kp = np.zeros(len(mem), KEYPOINT)
    kp["parentId"] = mem[:, 0]
    for v in range(len(features)):
        sel = mem[:, 0] == v
        kp["loc"][sel] = locs[v][mem[sel, 1]]
    return mm, kp


def triangulate(mm, kp, cameras, nview):
    """Bundle-range partitioned triangulation; every rank ends with the full cloud."""
    world, rank = _world()
    lo, hi = sd.bundle_range(len(mm), world, rank)
    sub = mm[lo:hi].copy()
    n = len(sub)
    pts = torch.zeros(0, dtype=torch.float32, device="cuda")
    if n:
        b_d, l_d = capi.generate_bundles(capi.to_dev(sub), capi.to_dev(kp), n, capi.to_dev(cameras), len(cameras), len(kp))
        pts, _, _ = capi.triangulate(l_d, b_d, n, nview=nview)
    if world == 1:
        return pts.view(-1, 3)
    parts = sd.all_gather_bytes(pts.view(torch.uint8).reshape(-1).contiguous())
    return torch.cat([p.view(torch.float32) for p in parts]).view(-1, 3)


def reconstruct(pixel_tensors_all, cameras, seed_features=None, epsilon=25.0, delta=5.0, mode=1, plans=None):
    """Full flow.  pixel_tensors_all: list of u8 CUDA tensors (only the owner rank's entries are used)."""
    world, rank = _world()
    num_images = len(pixel_tensors_all)
    mine = {v: pixel_tensors_all[v] for v in range(num_images) if sd.image_owner(v, world) == rank}
    feats = exchange_features(extract_features(mine, plans), num_images)
    pair_local = match_pairs(feats, cameras, seed_features, epsilon, delta, mode=mode)
    pair_all = exchange_pairs(pair_local, len(sd.pair_list(num_images)))
    mm, kp = build_match_set(feats, pair_all)
    cloud = triangulate(mm, kp, cameras, nview=num_images > 2)
    return {"features": feats, "matches": mm, "keypoints": kp, "points": cloud}
