"""Multi-GPU sharding of the N-view flow (BASELINE.json config[3]).

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
Units are independent -- SIFT per image, matching per ordered pair (i < j), triangulation per bundle -- so the only
collectives are the two exchanges the flow really has:

  exchange 1  all-gather of the per-image feature arrays (F x 152 B), so every rank holds both sides of its pairs;
  exchange 2  all-gather of the per-pair validated `uint2_pair` arrays (16 B per match) before the merge
              (MatchFactory::generateMatchesExhaustive, src/MatchFactory.cu:943-1020), which then runs replicated and
              deterministic on every rank (ssrlcv_hip_merge_matches on the device; merge_matches below is the host form).

Arrays have different lengths per rank, so each exchange is one small size collective (with a single device-to-host
copy of the size vector) followed by every rank's bytes at their exact size -- one broadcast per rank into its segment
of a flat buffer, queued together (they overlap under gloo; RCCL serialises them on the communicator's stream, which costs
nothing at these sizes; round 3 padded an all-gather to the largest rank: the shares differ several-fold).
Payloads are tens to hundreds of MB: the feature exchange of eight 4096^2 views moves 8 x ~55 MB (353 k features x 152 B each), well under a millisecond per link on 7 x 153 GB/s
xGMI links.

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
    """Stage B without size information: pair p (index in pair_list order) -> rank p mod G."""
    return pair_index % world


def assign_pairs(num_features, world):
    """Stage B, balanced: the cost of pair (i, j) is nq * nt distance evaluations, known on every rank once the feature
    arrays are exchanged.  Longest-processing-time-first: pairs by descending cost (ties by pair index) each go to the
    least loaded rank (ties to the lowest rank) -- deterministic, so every rank derives the same table.  With six pairs
    on four ranks round-robin gives two ranks two pairs each whatever their sizes.  -> owner rank per pair index.
    The table is computed by the library (ssrlcv_assign_pairs_host, csrc/host_merge.cpp: host code, no GPU): one
    definition for this driver and the C++ one (host/Distributed.hpp)."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    V = len(num_features)
    if V < 2:
        return []
    nf = (ctypes.c_uint32 * V)(*[int(x) for x in num_features])
    owners = (ctypes.c_uint32 * (V * (V - 1) // 2))()
    rc = lib.ssrlcv_assign_pairs_host(ctypes.c_uint32(V), nf, ctypes.c_uint32(int(world)), owners)
    if rc != 0:
        raise ValueError("ssrlcv_assign_pairs_host: status %d" % rc)
    return [int(o) for o in owners]


def bundle_range(num_bundles, world, rank):
    """Stage C: contiguous range of bundles for this rank."""
    per = (num_bundles + world - 1) // world
    lo = min(rank * per, num_bundles)
    return lo, min(lo + per, num_bundles)


def _comm_device(tensor_device, group=None):
    """Collectives run on the tensors' own device under RCCL; gloo (CPU tests, and the 2-ranks-on-one-GPU test) stages
    CUDA tensors through the host."""
    return torch.device("cpu") if dist.get_backend(group) == "gloo" else tensor_device


def all_reduce_sum(t, group=None):
    """In-place sum all-reduce that also works for CUDA tensors under gloo."""
    cd = _comm_device(t.device, group)
    if cd == t.device:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    else:
        h = t.to(cd)
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    return t


def exchange_mode():
    """How the variable-length exchanges move their bytes (SSRLCV_EXCHANGE, read at every exchange so that a bench run can
    A/B the two on the same communicator):
      "bcast"      (default) every rank's segment at its exact size: one broadcast per rank, queued together;
      "allgather"  ONE all-gather of segments padded to the largest rank's (the collective the north star names:
                   ncclAllGather; more bytes on the wire when the shares differ, one launch instead of world-many).
    Same result either way (tests/test_dist_gloo.py runs both)."""
    import os
    m = os.environ.get("SSRLCV_EXCHANGE", "bcast")
    if m not in ("bcast", "allgather"):
        raise ValueError("SSRLCV_EXCHANGE must be 'bcast' or 'allgather', not %r" % m)
    return m


def _gather_segments(segments, flat, group=None):
    """Fills `flat` (1-D uint8 on the backend's device): segment r = flat[offsets[r] : offsets[r] + sizes[r]] comes from
    rank r.  Default: exact sizes on the wire -- one broadcast per rank, queued together -- instead of an all-gather padded
    to the largest rank: the shares differ several-fold (a rank's features, or its LPT-assigned pairs).
    SSRLCV_EXCHANGE=allgather: one padded all-gather (see exchange_mode)."""
    if exchange_mode() == "allgather":
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        pad = max(max(size for _, size in segments), 1)
        pad = (pad + 15) // 16 * 16
        send = torch.zeros(pad, dtype=torch.uint8, device=flat.device)
        off, size = segments[rank]
        send[:size] = flat[off: off + size]
        recv = torch.empty(world * pad, dtype=torch.uint8, device=flat.device)
        dist.all_gather_into_tensor(recv, send, group=group)
        for r, (off, size) in enumerate(segments):
            if size and r != rank:
                flat[off: off + size] = recv[r * pad: r * pad + size]
        return
    works = []
    for r, (off, size) in enumerate(segments):
        if size:
            works.append(dist.broadcast(flat[off: off + size], src=r if group is None else dist.get_global_rank(group, r),
                                        group=group, async_op=True))
    for w in works:
        w.wait()


def all_gather_bytes(local, group=None):
    """All-gather of variable-length uint8 tensors.  `local` is a 1-D uint8 tensor on the backend's device (CUDA for
    nccl, CPU for gloo).  Returns a list with one tensor per rank.  One count all-gather (a single D2H copy of the
    world-sized count vector), then every rank's bytes at their exact size (no padding to the largest rank)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    home = local.device
    dev = _comm_device(home, group)
    local = local.to(dev)
    n = torch.tensor([local.numel()], dtype=torch.int64, device=dev)
    counts_t = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts_t, n, group=group)
    counts = counts_t.tolist()
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    flat = torch.empty(max(offs[-1], 1), dtype=torch.uint8, device=dev)
    flat[offs[rank]: offs[rank] + counts[rank]] = local
    _gather_segments([(offs[r], counts[r]) for r in range(world)], flat, group)
    flat = flat.to(home)
    return [flat[offs[r]: offs[r] + counts[r]] for r in range(world)]


def exchange_keyed(local_items, num_keys, owner_fn, group=None):
    """Generic keyed exchange: `local_items` maps key -> 1-D uint8 tensor for the keys this rank owns
    (owner_fn(key, world) == rank).  Every rank returns the full list [tensor for key 0, 1, ...].
    One all-reduce of the key sizes + every rank's bytes at their exact size (the keys of a rank concatenated into its segment
    of a flat buffer, one broadcast per rank: _gather_segments; nothing is padded to the largest rank)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = [k for k in range(num_keys) if owner_fn(k, world) == rank]
    dev = None
    for k in mine:
        dev = local_items[k].device
        break
    if dev is None:
        # a rank that owns no key still returns everybody's items where the others keep theirs: on its GPU when it has one
        # (also under gloo, which only stages through the host), on the CPU in the device-less tests
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    home = dev
    dev = _comm_device(home, group)
    # sizes of every key in one all-reduce (each rank fills in the keys it owns): one collective + one D2H copy
    sizes = torch.zeros(num_keys, dtype=torch.int64, device=dev)
    for k in mine:
        sizes[k] = local_items[k].numel()
    dist.all_reduce(sizes, op=dist.ReduceOp.SUM, group=group)
    sizes = sizes.tolist()
    per_rank = [sum(sizes[k] for k in range(num_keys) if owner_fn(k, world) == r) for r in range(world)]
    offs = [0]
    for c in per_rank:
        offs.append(offs[-1] + c)
    flat = torch.empty(max(offs[-1], 1), dtype=torch.uint8, device=dev)
    off = offs[rank]
    for k in mine:
        flat[off: off + sizes[k]] = local_items[k].reshape(-1).to(dev)
        off += sizes[k]
    _gather_segments([(offs[r], per_rank[r]) for r in range(world)], flat, group)   # exact sizes, no padding
    flat = flat.to(home)
    out = [None] * num_keys
    for r in range(world):
        off = offs[r]
        for k in range(num_keys):
            if owner_fn(k, world) == r:
                out[k] = flat[off: off + sizes[k]]
                off += sizes[k]
    return out


def merge_matches(num_features, pair_tensors):
    """Replicated host merge of the all-gathered uint2_pair arrays -> (MultiMatch array, members (image, feature))."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    counts = np.array([t.numel() // 16 for t in pair_tensors], np.uint32)
    live = [t.reshape(-1) for t in pair_tensors if t.numel()]
    # one device-side concatenation and one D2H copy (not one small copy per pair)
    allp = torch.cat(live).cpu().numpy() if live else np.zeros(0, np.uint8)
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
