"""N-view reconstruction flow (BASELINE.json configs[1..4]) over the HIP C ABI, single- or multi-GPU.

Mirrors src/Pipeline.cu doFeatureGeneration -> doFeatureMatching -> doTriangulation (+ the bundle-adjustment error
sweep of doBundleAdjust) with the sharding of ssrlcv_amd/dist.py:

  stage A   SIFT per image on its owner rank                         exchange 1: all-gather of the feature arrays
  stage B   pair matching on the pair's owner rank (pairs balanced   exchange 2: all-gather of the uint2_pair arrays
            by nq * nt, all queued, one synchronisation)
  merge     generateMatchesExhaustive's merge on the device, replicated (deterministic: csrc/merge.hip); KeyPoint table
            gathered on the device
  stage C   bundle-range triangulation                                exchange 3: all-gather of the cloud
  BA sweep  the K finite-difference evaluations of f(cameras) over this rank's bundle range of the first image pair
            (calculateImageGradient / Hessian, src/PointCloudFactory.cu:1059-1504)   all-reduce(sum) of the K sums

With world size 1 (no process group) every exchange is the identity.  Python here is plumbing (device buffers,
torch.distributed); all compute is libssrlcv_hip.so.  A `Workspace` keeps the per-rank plans, matcher workspaces and
seed descriptors between calls: creating and freeing them per pair / per image was a third of a step.
"""
import os
import sys
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
        self._plan_bytes = {}
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

    def plan_bytes(self, w, h):
        """Workspace + feature buffer a plan of this size holds (from a layout-only plan: no device memory)."""
        key = (w, h)
        if key not in self._plan_bytes:
            probe = capi.SiftPlan(w, h, alloc=False)
            self._plan_bytes[key] = probe.workspace_bytes + probe.max_features * FEATURE_BYTES
        return self._plan_bytes[key]

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


def extract_features(pixel_tensors, ws=None, plan_budget_bytes=None):
    """SIFT on the images this rank owns.  pixel_tensors: {image index: u8 CUDA tensor (H, W)} -> {index: feature bytes}.
    While the plans' workspaces fit the budget (default: 40 % of the device's memory) every owned image gets its own plan
    and the extracts are queued back to back and synchronised once; past it (eight 8192^2 views on one GPU need 8 x 22 GB)
    one plan per image size is reused: extract, copy the features out, next image."""
    ws = ws or Workspace()
    if plan_budget_bytes is None:
        plan_budget_bytes = int(0.4 * torch.cuda.get_device_properties(torch.cuda.current_device()).total_memory)
    items = list(pixel_tensors.items())
    need = 0
    for k, (v, pix) in enumerate(items):
        h, w = pix.shape
        need += ws.plan_bytes(w, h)
    out = {}
    if need <= plan_budget_bytes:
        plans = {}
        for k, (v, pix) in enumerate(items):
            h, w = pix.shape
            plans[v] = ws.plan(w, h, k)
            plans[v].extract(pix)
        for v, plan in plans.items():
            n = plan.count()   # synchronises; raises if a key-point list overflowed
            out[v] = plan.features[: n * FEATURE_BYTES].clone()
        return out
    for v, pix in items:
        h, w = pix.shape
        plan = ws.plan(w, h, 0)
        plan.extract(pix)
        n = plan.count()
        out[v] = plan.features[: n * FEATURE_BYTES].clone()
    return out


def exchange_features(local, num_images):
    world, _ = _world()
    if world == 1:
        return [local[v] for v in range(num_images)]
    return sd.exchange_keyed(local, num_images, sd.image_owner)


def match_pairs(features, cameras, seed_features=None, epsilon=25.0, delta=5.0, rel=0.6, absolute=200.0 * 200.0, mode=1,
                ws=None, owners=None):
    """Exhaustive matching (generateMatchesExhaustive; GEO_ORBIT double-constrained for mode 1, brute force for mode 0)
    of the pairs this rank owns (`owners`: rank per pair index, default dist.assign_pairs).  features: list of uint8 CUDA
    tensors (all images).  Every pair is queued -- match, validation / compaction with the count left on the device --
    and the stream is synchronised once for all the counts.  Returns {pair index: validated uint2_pair bytes}."""
    ws = ws or Workspace()
    world, rank = _world()
    num_images = len(features)
    pairs = sd.pair_list(num_images)
    if owners is None:
        owners = sd.assign_pairs([f.numel() // FEATURE_BYTES for f in features], world)
    seed_d = ws.seed_features(seed_features)
    mine = [p for p in range(len(pairs)) if owners[p] == rank]
    counts = torch.zeros(max(len(mine), 1), dtype=torch.int32, device="cuda")
    bufs = {}
    seed_cache = {}
    shapes = []
    for p in mine:
        qi, ti = pairs[p]
        nq, nt = features[qi].numel() // FEATURE_BYTES, features[ti].numel() // FEATURE_BYTES
        shapes.append((nq, nt))
        if seed_features is not None:
            shapes.append((nq, len(seed_features)))
    mws = ws.matcher(shapes, 16)[0] if shapes else None
    for k, p in enumerate(mine):
        qi, ti = pairs[p]
        nq, nt = features[qi].numel() // FEATURE_BYTES, features[ti].numel() // FEATURE_BYTES
        sdist = None
        if seed_d is not None:
            if qi not in seed_cache:  # recomputed per query image upstream (src/MatchFactory.cu:925)
                seed_cache[qi] = capi.seed_distances(features[qi], nq, seed_d, len(seed_features), workspace=mws)
            sdist = seed_cache[qi]
        proj = capi.projection_matrix(cameras[ti:ti + 1]) if mode == 1 else None
        params = capi.make_match_params(mode, qi, ti, epsilon, delta, rel, absolute,
                                        cameras[qi:qi + 1] if mode == 1 else None, proj)
        bufs[p] = capi.dev_bytes(nq * 16)
        capi.match(features[qi], nq, features[ti], nt, params, capi.OUT_UINT2_PAIR, seed_d=sdist, workspace=mws, out=bufs[p])
        capi.compact_matches_async(capi.OUT_UINT2_PAIR, bufs[p], nq, mws, counts[k:k + 1])
    n = counts.cpu().tolist()  # the one synchronisation
    return {p: bufs[p][: n[k] * 16] for k, p in enumerate(mine)}


def exchange_pairs(local, num_pairs, owners=None):
    world, _ = _world()
    if world == 1:
        return [local[p] for p in range(num_pairs)]
    owner_fn = sd.pair_owner if owners is None else (lambda p, _world_size: owners[p])
    return sd.exchange_keyed(local, num_pairs, owner_fn)


def _pinned_copy(t_d, nbytes, slot):
    """D2H through a pinned staging buffer kept between calls (pageable, 16 MB took 5.6 ms) -> uint8 numpy view of the
    staging buffer (valid until the next call with the same slot)."""
    stages = build_match_set.__dict__.setdefault("_stages", {})
    if slot not in stages or stages[slot].numel() < nbytes:
        stages[slot] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, pin_memory=True)
    stage = stages[slot][:nbytes]
    stage.copy_(t_d[:nbytes])
    return stage.numpy()


def _pool_probe():
    """How sys.getrefcount reads for a pooled array nobody else refers to, measured with the very loop _result_buffer runs
    (the pool's list, the loop variable, getrefcount's argument: 3 on CPython 3.10; interpreters that borrow references
    read lower), and checked: one outstanding view must read exactly one more.  None = the count cannot be trusted on this
    interpreter, and the pool is not used (every result gets a fresh array)."""
    pool = [np.empty(16, np.uint8)]
    free = held = None
    for buf in pool:
        free = sys.getrefcount(buf)
    view = pool[0][:8].view(np.uint16)   # what a caller holds: a view whose base is the pooled buffer
    for buf in pool:
        held = sys.getrefcount(buf)
    del view
    for buf in pool:
        again = sys.getrefcount(buf)
    return free if (held == free + 1 and again == free) else None


_POOL_FREE_COUNT = _pool_probe()


def _result_buffer(nbytes, slot):
    """A host byte array for a result the caller will own.  Fresh arrays of this size (12 MB of key points per step of the
    4 x 4096^2 flow) are mmap'ed and page-faulted by every call, and now and then that stalls for 25-30 ms (seen in bench.py's
    N-view leg: one `merge` stage in five).  So the arrays are pooled per slot and handed out again once nobody outside the
    pool refers to them any more: a result array, and every slice or view a caller takes of it, is a view whose base is the
    pooled buffer, so the buffer's reference count tells -- against the free reading calibrated by _pool_probe() on this
    interpreter, not a constant."""
    if _POOL_FREE_COUNT is None:
        return np.empty(max(nbytes, 1 << 16), np.uint8)
    pool = build_match_set.__dict__.setdefault("_results", {}).setdefault(slot, [])
    for buf in pool:
        if buf.size >= nbytes and sys.getrefcount(buf) == _POOL_FREE_COUNT:
            return buf
    buf = np.empty(max(nbytes, 1 << 16), np.uint8)
    if len(pool) >= 4:
        pool.pop(0)
    pool.append(buf)
    return buf


def _host_records(t_d, count, dtype, slot):
    """`count` records of a device byte tensor as a structured numpy array the caller owns.  The copy out of the staging
    buffer is a plain byte copy: `view(dtype).copy()` on a structured dtype went element by element and took 5 of the
    merge stage's 7 ms for 0.7 M key points."""
    nbytes = count * dtype.itemsize
    raw = _result_buffer(nbytes, slot)[:nbytes]
    if count:
        np.copyto(raw, _pinned_copy(t_d, nbytes, slot))
    return raw.view(dtype)


def build_match_set(features, pair_tensors, dev=None):
    """Replicated merge -> (MultiMatch numpy, KeyPoint numpy).  The merge (src/MatchFactory.cu:943-1020) and the KeyPoint
    table (image, location of every member, :1007-1020) are built on the device (csrc/merge.hip, every rank from the same
    all-gathered pairs: deterministic); `dev` (a dict) receives the device copies so that the triangulation does not upload
    them again.  SSRLCV_MERGE_HOST=1 (or more than 32 images) takes the host merge (csrc/host_merge.cpp) instead."""
    num_features = [f.numel() // FEATURE_BYTES for f in features]
    counts = [t.numel() // 16 for t in pair_tensors]
    use_host = bool(os.environ.get("SSRLCV_MERGE_HOST")) or len(features) > 32
    if not use_host:
        live = [t.reshape(-1) for t in pair_tensors if t.numel()]
        pairs_d = torch.cat(live) if live else torch.zeros(16, dtype=torch.uint8, device="cuda")
        try:
            mm_d, mem_d, n_mm, n_mem, rounds, build_match_set._ws = capi.merge_matches_device(
                num_features, counts, pairs_d, getattr(build_match_set, "_ws", None))
        except capi.MalformedPairList:
            use_host = True   # e.g. a query matched twice in one pair: only upstream's host walk defines a result for it
    if use_host:
        mm, mem = sd.merge_matches(num_features, pair_tensors)
        kp = np.zeros(len(mem), KEYPOINT)
        if len(mem):
            mem_d = capi.to_dev(np.ascontiguousarray(mem, np.uint32))
            kp_d = capi.keypoints_from_members(mem_d, len(mem), features)
            kp = _host_records(kp_d, len(mem), KEYPOINT, "kp")   # padding bytes are zeros (k_keypoints_from_members)
            if dev is not None:
                dev["keypoints"] = kp_d
        return mm, kp
    mm = np.zeros(0, MULTIMATCH)
    kp = np.zeros(0, KEYPOINT)
    if n_mem:
        kp_d = capi.keypoints_from_members(mem_d, n_mem, features)
        mm = _host_records(mm_d, n_mm, MULTIMATCH, "mm")
        kp = _host_records(kp_d, n_mem, KEYPOINT, "kp")   # padding bytes are zeros (k_keypoints_from_members)
        if dev is not None:
            dev["keypoints"] = kp_d
            dev["matches"] = mm_d
    return mm, kp


def triangulate(mm, kp, cameras, nview, pushbroom=None, dev=None):
    """Bundle-range partitioned triangulation; every rank ends with the full cloud.  `pushbroom`: PushbroomCamera array
    (config[4]): bundles then come from generatePushbroomBundle (src/PointCloudFactory.cu:875-903).  `dev`: device copies
    left by build_match_set."""
    world, rank = _world()
    lo, hi = sd.bundle_range(len(mm), world, rank)
    n = hi - lo
    pts = torch.zeros(0, dtype=torch.float32, device="cuda")
    if n:
        kp_d = dev["keypoints"] if dev and "keypoints" in dev else capi.to_dev(kp)
        sub_d = dev["matches"][8 * lo: 8 * hi] if dev and "matches" in dev else capi.to_dev(mm[lo:hi].copy())
        if pushbroom is not None:
            b_d, l_d = capi.generate_pushbroom_bundles(sub_d, kp_d, n, capi.to_dev(pushbroom),
                                                       len(pushbroom), len(kp))
        else:
            b_d, l_d = capi.generate_bundles(sub_d, kp_d, n, capi.to_dev(cameras), len(cameras), len(kp))
        pts, _, _ = capi.triangulate(l_d, b_d, n, nview=nview)
    if world == 1:
        return pts.view(-1, 3)
    parts = sd.all_gather_bytes(pts.view(torch.uint8).reshape(-1).contiguous())
    return torch.cat([p.view(torch.float32) for p in parts]).view(-1, 3)


def filter_match_set(mm_d, kp_d, n, n_kp, cameras, kind, cutoff=None, sigma=3.0, sample_size=0.1, pushbroom=None):
    """PointCloudFactory::linearCutoffFilter (kind "linear", src/PointCloudFactory.cu:3500-3644) or
    deterministicStatisticalFilter (kind "statistical", :3070-3275) on a MatchSet that lives on the device (MultiMatch /
    KeyPoint byte tensors): bundles, the error triangulation, the statistical cutoff, the cutoff triangulation and the
    rebuild of the MatchSet are queued back to back, the two counts come back in one small copy.
    -> (mm_d, kp_d, n, n_kp) after the filter (the inputs themselves when upstream would leave the MatchSet alone)."""
    if n == 0:
        return mm_d, kp_d, n, n_kp
    nview = len(cameras if pushbroom is None else pushbroom) > 2
    if pushbroom is not None:
        b_d, l_d = capi.generate_pushbroom_bundles(mm_d, kp_d, n, capi.to_dev(pushbroom), len(pushbroom), n_kp)
    else:
        b_d, l_d = capi.generate_bundles(mm_d, kp_d, n, capi.to_dev(cameras), len(cameras), n_kp)
    if kind == "linear":
        if cutoff < 0.0:
            return mm_d, kp_d, n, n_kp
        capi.triangulate(l_d, b_d, n, nview=nview, want_errors=True, cutoff=float(cutoff))
        only_if_bad = True
    else:
        if sample_size > 1.0 or sample_size < 0.0:
            return mm_d, kp_d, n, n_kp
        _, err_d, _ = capi.triangulate(l_d, b_d, n, nview=nview, want_errors=True, cutoff=0.0)
        cut_d = capi.error_sample_cutoff(err_d, n, int(1 / sample_size), sigma)
        capi.triangulate(l_d, b_d, n, nview=nview, want_errors=True, cutoff=cut_d)
        only_if_bad = nview
    mm_o, kp_o, counts = capi.filter_matchset(b_d, kp_d, n, n_kp)
    kept, kept_kp, _ = [int(x) for x in counts.cpu().tolist()]
    if (only_if_bad and kept == n) or kept == 0 or (nview and kept_kp == 0):
        return mm_d, kp_d, n, n_kp   # nothing to remove, or "filtering is too aggressive": the MatchSet stays as it is
    return mm_o[: 8 * kept], kp_o[: 16 * kept_kp], kept, kept_kp


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


def _ba_subset_device(dev, n_mm, pair, world, rank):
    """The 2-view bundles of `pair` out of the device copy of the MatchSet as a 2-camera MatchSet on the device
    (ssrlcv_hip_select_pair_bundles: one library pass, the count is the only thing that comes back), and this rank's share
    of them: the MultiMatch entries index the shared KeyPoint array, so a share is a slice of the MultiMatch array.
    -> (MultiMatch bytes of the share, KeyPoint bytes, bundles of this rank, bundles of the pair)"""
    n_kp = dev["keypoints"].numel() // 16
    sub_mm, sub_kp, count = capi.select_pair_bundles(dev["matches"], dev["keypoints"], n_mm, n_kp, pair[0], pair[1])
    total = int(count.item())
    lo, hi = sd.bundle_range(total, world, rank)
    return sub_mm[8 * lo: 8 * hi], sub_kp, hi - lo, total


def ba_error_sweep(mm, kp, cameras, pair=(0, 1), params=None, dev=None):
    """f(params_k) = sum of squared skew-line gaps of the 2-view bundles of `pair`, for all K parameter sets: this rank's
    bundle range in one launch (ssrlcv_hip_ba_sweep2), then all-reduce(sum) over the ranks.  Returns a K-vector on the
    device (identical on every rank up to the float all-reduce).  `dev`: the device copies of the MatchSet left by
    build_match_set / apply_filters; with them the pair's bundles are selected on the device."""
    world, rank = _world()
    if dev and "matches" in dev and "keypoints" in dev and len(mm):
        cams2 = cameras[[pair[0], pair[1]]].copy()
        if params is None:
            params = ba_parameter_sets(cams2)
        K = len(params)
        sub_mm_d, sub_kp_d, n, total = _ba_subset_device(dev, len(mm), pair, world, rank)
        sums = torch.zeros(K, dtype=torch.float32, device="cuda")
        if n:
            sums = capi.ba_sweep2(sub_mm_d, sub_kp_d, n, capi.to_dev(cams2), 2,
                                  torch.from_numpy(np.ascontiguousarray(params, np.float32)).cuda(), K)
        if world > 1:
            sd.all_reduce_sum(sums)
        return sums, total
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


REFERENCE_FILTERS = "reference"   # doFiltering's own sequence (src/Pipeline.cu:305-340)


def apply_filters(mm, kp, dev, cameras, filters, pushbroom=None):
    """The filtering stage on the device copies of the MatchSet (dev["matches"], dev["keypoints"]; uploaded if the merge ran
    on the host).  filters: REFERENCE_FILTERS = what doFiltering runs -- the 100 km linear cutoff then the 3 sigma / 10 %
    statistical filter for two views, the statistical filter alone otherwise -- or a list of ("linear", cutoff) /
    ("statistical", sigma, sample_size) steps.  Every rank filters the same replicated MatchSet (deterministic).
    -> (MultiMatch numpy, KeyPoint numpy) after the filters; dev holds the filtered device copies."""
    views = len(cameras if pushbroom is None else pushbroom)
    if filters == REFERENCE_FILTERS:
        filters = ([("linear", 100.0)] if views == 2 else []) + [("statistical", 3.0, 0.1)]
    n, n_kp = len(mm), len(kp)
    if n == 0 or not filters:
        return mm, kp
    mm_d = dev["matches"] if "matches" in dev else capi.to_dev(mm)
    kp_d = dev["keypoints"] if "keypoints" in dev else capi.to_dev(kp)
    changed = False
    for f in filters:
        if f[0] == "linear":
            out = filter_match_set(mm_d, kp_d, n, n_kp, cameras, "linear", cutoff=f[1], pushbroom=pushbroom)
        else:
            out = filter_match_set(mm_d, kp_d, n, n_kp, cameras, "statistical", sigma=f[1], sample_size=f[2], pushbroom=pushbroom)
        changed = changed or out[2] != n
        mm_d, kp_d, n, n_kp = out
    dev["matches"], dev["keypoints"] = mm_d, kp_d
    if changed:
        mm = _host_records(mm_d, n, MULTIMATCH, "mm")
        kp = _host_records(kp_d, n_kp, KEYPOINT, "kp")
    return mm, kp


def reconstruct(pixel_tensors_all, cameras, seed_features=None, epsilon=25.0, delta=5.0, mode=1, ws=None, pushbroom=None,
                ba=False, filters=None):
    """Full flow.  pixel_tensors_all: list of u8 CUDA tensors (only the owner rank's entries are used).  `ws`: a
    Workspace kept by the caller between calls; its `times` dict accumulates the wall time of every stage.
    `filters` (None, REFERENCE_FILTERS or a list, see apply_filters): the filtering stage between the MatchSet and the
    cloud, like doFiltering upstream; the result's matches / keypoints / points are then the filtered ones and
    "matches_unfiltered" says how many multi-matches went in."""
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
    owners = sd.assign_pairs([f.numel() // FEATURE_BYTES for f in feats], world)
    pair_local = match_pairs(feats, cameras, seed_features, epsilon, delta, mode=mode, ws=ws, owners=owners)
    ws.tick("match", t)
    t = time.perf_counter()
    pair_all = exchange_pairs(pair_local, len(sd.pair_list(num_images)), owners)
    ws.tick("exchange_pairs", t)
    t = time.perf_counter()
    dev = {}
    mm, kp = build_match_set(feats, pair_all, dev)
    ws.tick("merge", t)
    unfiltered = len(mm)
    if filters:
        t = time.perf_counter()
        mm, kp = apply_filters(mm, kp, dev, cameras, filters, pushbroom)
        ws.tick("filter", t)
    t = time.perf_counter()
    cloud = triangulate(mm, kp, cameras, nview=num_images > 2, pushbroom=pushbroom, dev=dev)
    torch.cuda.synchronize()
    ws.tick("triangulate", t)
    out = {"features": feats, "pairs": pair_all, "matches": mm, "keypoints": kp, "points": cloud, "matches_unfiltered": unfiltered,
           "device": dev}   # device copies of the MatchSet ("matches", "keypoints"), e.g. for further apply_filters steps
    if ba and pushbroom is None:
        t = time.perf_counter()
        out["ba_sums"], out["ba_bundles"] = ba_error_sweep(mm, kp, cameras, dev=dev)
        torch.cuda.synchronize()
        ws.tick("ba_sweep", t)
    return out
