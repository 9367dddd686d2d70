"""Multi-GPU sharding of the N-view flow (BASELINE.json config[3]).

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
Units are independent -- SIFT per image, matching per ordered pair (i < j), triangulation per bundle -- so the only
collectives are the two exchanges the flow really has:

  exchange 1  all-gather of the per-image feature arrays (F x 152 B), so every rank holds both sides of its pairs;
  exchange 2  all-gather of the per-pair validated `uint2_pair` arrays (16 B per match) before the host merge
              (MatchFactory::generateMatchesExhaustive, src/MatchFactory.cu:943-1020), which then runs replicated and
              deterministic on every rank (ssrlcv_merge_matches_host).

Arrays have different lengths per rank, so each exchange is a count all-gather followed by one padded all-gather
(payloads are tens of MB at most: latency-, not bandwidth-bound on 7 x 153 GB/s xGMI links).

This module holds only the sharding / exchange logic; compute calls go through ssrlcv_amd.capi (HIP C ABI).
"""
import numpy as np
import torch
import torch.distributed as dist


def image_owner(image, world):
    """Stage A: image v -> rank v mod G."""
    return image % world


def pair_list(num_images):
    """All pairs i < j in the reference's iteration order (src/MatchFactory.cu:924-936)."""
    return [(i, j) for i in range(num_images - 1) for j in range(i + 1, num_images)]


def pair_owner(pair_index, world):
    """Stage B: pair p (index in pair_list order) -> rank p mod G."""
    return pair_index % world


def bundle_range(num_bundles, world, rank):
    """Stage C: contiguous range of bundles for this rank."""
    per = (num_bundles + world - 1) // world
    lo = min(rank * per, num_bundles)
    return lo, min(lo + per, num_bundles)


def all_gather_bytes(local, group=None):
    """All-gather of variable-length uint8 tensors.  `local` is a 1-D uint8 tensor on the backend's device (CUDA for
    nccl, CPU for gloo).  Returns a list with one tensor per rank."""
    world = dist.get_world_size(group)
    dev = local.device
    n = torch.tensor([local.numel()], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(max(counts), 1)
    padded = torch.zeros(mx, dtype=torch.uint8, device=dev)
    padded[: local.numel()] = local
    out = [torch.empty(mx, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    return [o[:c] for o, c in zip(out, counts)]


def exchange_keyed(local_items, num_keys, owner_fn, group=None):
    """Generic keyed exchange: `local_items` maps key -> 1-D uint8 tensor for the keys this rank owns
    (owner_fn(key, world) == rank).  Every rank returns the full list [tensor for key 0, 1, ...].
    One count all-gather + one padded payload all-gather per call (keys of a rank are concatenated)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = [k for k in range(num_keys) if owner_fn(k, world) == rank]
    dev = None
    for k in mine:
        dev = local_items[k].device
        break
    if dev is None:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    sizes = torch.tensor([local_items[k].numel() for k in mine], dtype=torch.int64, device=dev)
    size_bytes = sizes.view(torch.uint8) if len(mine) else torch.zeros(0, dtype=torch.uint8, device=dev)
    all_sizes = all_gather_bytes(size_bytes.contiguous(), group)
    payload = torch.cat([local_items[k].reshape(-1) for k in mine]) if mine else torch.zeros(0, dtype=torch.uint8, device=dev)
    all_payload = all_gather_bytes(payload.contiguous(), group)
    out = [None] * num_keys
    for r in range(world):
        keys_r = [k for k in range(num_keys) if owner_fn(k, world) == r]
        sz = all_sizes[r].view(torch.int64).tolist() if keys_r else []
        off = 0
        for k, s in zip(keys_r, sz):
            out[k] = all_payload[r][off: off + s]
            off += s
    return out


def merge_matches(num_features, pair_tensors):
    """Replicated host merge of the all-gathered uint2_pair arrays -> (MultiMatch array, members (image, feature))."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    counts = np.array([t.numel() // 16 for t in pair_tensors], np.uint32)
    allp = np.concatenate([t.cpu().numpy().reshape(-1) for t in pair_tensors]) if len(pair_tensors) else np.zeros(0, np.uint8)
    allp = np.ascontiguousarray(allp)
    nf = np.array(num_features, np.uint32)
    mm_p, mem_p = ctypes.c_void_p(), ctypes.c_void_p()
    nmm, nmem = ctypes.c_uint32(), ctypes.c_uint32()
    rc = lib.ssrlcv_merge_matches_host(ctypes.c_uint32(len(nf)), nf.ctypes.data_as(ctypes.c_void_p),
                                       ctypes.c_uint32(len(counts)), counts.ctypes.data_as(ctypes.c_void_p),
                                       allp.ctypes.data_as(ctypes.c_void_p), ctypes.byref(mm_p), ctypes.byref(mem_p),
                                       ctypes.byref(nmm), ctypes.byref(nmem))
    if rc != 0:
        raise RuntimeError("ssrlcv_merge_matches_host failed: %d" % rc)
    mm = np.ctypeslib.as_array(ctypes.cast(mm_p, ctypes.POINTER(ctypes.c_uint8)), shape=(max(nmm.value, 1) * 8,))[: nmm.value * 8].copy()
    mem = np.ctypeslib.as_array(ctypes.cast(mem_p, ctypes.POINTER(ctypes.c_uint32)), shape=(max(nmem.value, 1), 2))[: nmem.value].copy()
    lib.ssrlcv_host_free(mm_p)
    lib.ssrlcv_host_free(mem_p)
    mm = mm.view(np.dtype([("numKeyPoints", "<u4"), ("index", "<i4")]))
    return mm, mem
