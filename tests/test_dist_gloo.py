"""gloo tests of the multi-GPU exchange logic (run on CPU, world sizes 2, 3 and 4): variable-length all-gather, keyed
exchange of per-image feature arrays and per-pair uint2_pair arrays (round-robin owners and the cost-balanced table of
the flow), and the replicated deterministic merge, checked against the single-process result and the oracle's merge --
four views, and the eight views / 28 pairs of the 8-GPU strong-scaling leg (bench.py `nview`) with feature arrays of the
size 1024^2 views produce, on a rank count that divides neither the images nor the pairs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _synthetic_pairs(num_images, num_features, seed):
    """Consistent random matches between images: a hidden 'track id' per feature makes some chains transitive."""
    rng = np.random.default_rng(seed)
    tracks = [rng.permutation(4 * nf)[:nf] for nf in num_features]
    lists = []
    from ssrlcv_amd import dist as sd
    for (i, j) in sd.pair_list(num_images):
        lookup = {t: idx for idx, t in enumerate(tracks[j])}
        rows = []
        for f, t in enumerate(tracks[i]):
            if t in lookup and rng.random() < 0.9:
                rows.append((i, f, j, lookup[t]))
            elif rng.random() < 0.05:
                rows.append((i, f, j, int(rng.integers(0, num_features[j]))))  # inconsistent match
        arr = np.array(rows, np.uint32).reshape(-1, 4) if rows else np.zeros((0, 4), np.uint32)
        lists.append(arr)
    return lists


def _feature_counts(num_images):
    """four small views (the round-1 case) or what 1024^2 scene views give: 60-95 k features each"""
    if num_images == 4:
        return [300, 280, 310, 290]
    rng = np.random.default_rng(num_images)
    return [int(x) for x in rng.integers(60000, 95000, num_images)]


def _worker(rank, world, port, tmp, num_images, mode="bcast"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SSRLCV_EXCHANGE"] = mode   # per-rank exact-size broadcasts, or one padded all-gather (dist.exchange_mode)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssrlcv_amd import dist as sd
    try:
        # variable-length all-gather, including an empty contribution
        local = torch.arange(rank * 5, dtype=torch.uint8)
        got = sd.all_gather_bytes(local)
        assert [g.numel() for g in got] == [r * 5 for r in range(world)]
        assert all(torch.equal(g, torch.arange(r * 5, dtype=torch.uint8)) for r, g in enumerate(got))
        # exchange 1: per-image feature arrays owned by image % world
        num_features = _feature_counts(num_images)
        feats = {}
        for v in range(num_images):
            if sd.image_owner(v, world) == rank:
                rng = np.random.default_rng(100 + v)
                feats[v] = torch.from_numpy(rng.integers(0, 256, num_features[v] * 152, dtype=np.uint8))
        allf = sd.exchange_keyed(feats, num_images, sd.image_owner)
        for v in range(num_images):
            rng = np.random.default_rng(100 + v)
            assert torch.equal(allf[v], torch.from_numpy(rng.integers(0, 256, num_features[v] * 152, dtype=np.uint8)))
        # exchange 2: per-pair uint2_pair arrays owned by pair index % world, then the replicated merge
        pair_lists = _synthetic_pairs(num_images, num_features, 7)
        mine = {p: torch.from_numpy(pair_lists[p].view(np.uint8).reshape(-1).copy())
                for p in range(len(pair_lists)) if sd.pair_owner(p, world) == rank}
        allp = sd.exchange_keyed(mine, len(pair_lists), sd.pair_owner)
        for p in range(len(pair_lists)):
            assert np.array_equal(allp[p].numpy().view(np.uint32).reshape(-1, 4), pair_lists[p])
        # ... and with the owners of the flow's stage B: the longest-processing-time-first table every rank derives from the
        # exchanged feature counts (dist.assign_pairs), a rank possibly owning no pair at all
        owners = sd.assign_pairs(num_features, world)
        assert len(owners) == len(pair_lists) and owners == sd.assign_pairs(list(num_features), world)
        mine = {p: torch.from_numpy(pair_lists[p].view(np.uint8).reshape(-1).copy())
                for p in range(len(pair_lists)) if owners[p] == rank}
        allq = sd.exchange_keyed(mine, len(pair_lists), lambda p, _w: owners[p])
        for p in range(len(pair_lists)):
            assert torch.equal(allq[p], allp[p])
        mm, mem = sd.merge_matches(num_features, allp)
        np.save(os.path.join(tmp, "mm_%d.npy" % rank), mm)
        np.save(os.path.join(tmp, "mem_%d.npy" % rank), mem)
        lo, hi = sd.bundle_range(len(mm), world, rank)
        assert 0 <= lo <= hi <= len(mm)
        # stage C: the ranges tile the bundles exactly; the cloud all-gather puts every rank's share at its place
        rng_all = [sd.bundle_range(len(mm), world, r) for r in range(world)]
        assert rng_all[0][0] == 0 and rng_all[-1][1] == len(mm) and all(a[1] == b[0] for a, b in zip(rng_all, rng_all[1:]))
        share = torch.arange(lo, hi, dtype=torch.int32).view(torch.uint8)
        cloud = torch.cat(sd.all_gather_bytes(share)).view(torch.int32)
        assert torch.equal(cloud, torch.arange(len(mm), dtype=torch.int32))
        # the BA sweep's all-reduce: K partial sums per rank
        part = torch.full((8,), float(rank + 1), dtype=torch.float32)
        assert torch.equal(sd.all_reduce_sum(part), torch.full((8,), world * (world + 1) / 2.0))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,num_images,mode", [(2, 4, "bcast"), (2, 8, "bcast"), (3, 8, "bcast"), (4, 8, "bcast"),
                                                   (2, 4, "allgather"), (3, 8, "allgather"), (4, 8, "allgather")])
def test_exchanges_and_replicated_merge_on_several_ranks(tmp_path, oracle_lib, world, num_images, mode):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), num_images, mode), nprocs=world, join=True)
    mms = [np.load(tmp_path / ("mm_%d.npy" % r)) for r in range(world)]
    mems = [np.load(tmp_path / ("mem_%d.npy" % r)) for r in range(world)]
    mm0, mem0 = mms[0], mems[0]
    for r in range(1, world):
        assert np.array_equal(mm0, mms[r]) and np.array_equal(mem0, mems[r])  # replicated merge is deterministic
    assert len(mm0) > 50 and (mm0["numKeyPoints"] >= 2).all() and (mm0["numKeyPoints"] <= num_images).all()
    # single-process product merge and the oracle's restatement agree with the sharded result
    from ssrlcv_amd import dist as sd
    num_features = _feature_counts(num_images)
    lists = _synthetic_pairs(num_images, num_features, 7)
    tens = [torch.from_numpy(l.view(np.uint8).reshape(-1).copy()) for l in lists]
    mm_s, mem_s = sd.merge_matches(num_features, tens)
    assert np.array_equal(mm_s, mm0) and np.array_equal(mem_s, mem0)
    if num_images == 4 or world == 3:   # (the oracle's walk is a single thread: once per view count is enough)
        omm, omem = H.oracle_merge(oracle_lib, num_features, [l.view(H.UINT2_PAIR).reshape(-1) for l in lists])
        assert np.array_equal(omm["numKeyPoints"], mm0["numKeyPoints"]) and np.array_equal(omm["index"], mm0["index"])
        assert np.array_equal(omem, mem0)


def test_merge_reproduces_reference_3view_fixture(oracle_lib, everest_oracle_features):
    """The library's host merge on the oracle's pair lists -> the reference's Pipeline3View MultiMatch fixture."""
    from ssrlcv_amd import dist as sd
    feats = everest_oracle_features
    seed, _ = H.load_seed_features()
    v = H.load_view("Pipeline3View")
    cams = v["cameras"]
    pair_lists = []
    for qi in range(2):
        sdist = H.oracle_seed_distances(oracle_lib, feats[qi], seed)
        for ti in range(qi + 1, 3):
            proj = H.oracle_projection(oracle_lib, cams[ti:ti + 1])
            pr = H.oracle_match_pairs(oracle_lib, 1, qi, feats[qi], ti, feats[ti], cams[qi:qi + 1], proj, 25.0, 5.0,
                                      sdist, 0.6, 200.0 * 200.0)
            pair_lists.append(pr[~(pr["a"] == pr["b"]).all(1)])
    tens = [torch.from_numpy(p.view(np.uint8).reshape(-1).copy()) for p in pair_lists]
    mm, mem = sd.merge_matches([len(f) for f in feats], tens)
    assert len(mm) == 21177
    assert np.array_equal(mm["numKeyPoints"], v["mm0"]["numKeyPoints"]) and np.array_equal(mm["index"], v["mm0"]["index"])
    assert np.array_equal(mem[:, 0].astype(np.int32), v["kp0"]["parentId"])


def test_pair_assignment_is_balanced_and_deterministic():
    """Stage B's longest-processing-time-first table (dist.assign_pairs): every rank derives the same owners from the
    feature counts, each pair has exactly one owner, and the heaviest rank carries no more than the lightest plus the
    largest single pair (round-robin on six pairs over four ranks gives two ranks two pairs each whatever their sizes)."""
    from ssrlcv_amd import dist as sd
    nf = [300000, 120000, 410000, 90000]
    for world in (1, 2, 3, 4, 8):
        owners = sd.assign_pairs(nf, world)
        assert owners == sd.assign_pairs(list(nf), world) and len(owners) == 6 and all(0 <= r < world for r in owners)
        pairs = sd.pair_list(len(nf))
        cost = [nf[i] * nf[j] for i, j in pairs]
        load = [sum(c for c, r in zip(cost, owners) if r == k) for k in range(world)]
        assert max(load) - min(load) <= max(cost)
    # equal sizes on four ranks: the two ranks that take a second pair are decided by the tie rule, not by chance
    assert sd.assign_pairs([1000] * 4, 4) == [0, 1, 2, 3, 0, 1]


def test_pair_table_is_longest_processing_time_first():
    from ssrlcv_amd import dist as sd
    """ssrlcv_assign_pairs_host (the one definition behind dist.assign_pairs and host/Distributed.hpp) against the rule
    written out: pairs by descending cost nq x nt (ties by pair index), each to the least loaded rank (ties to the lowest)."""
    rng = np.random.default_rng(3)
    for trial in range(200):
        V = int(rng.integers(2, 10))
        world = int(rng.integers(1, 9))
        nf = [int(x) for x in rng.integers(0, 5, V)] if trial % 3 == 0 else [int(x) for x in rng.integers(1000, 3000000, V)]
        pairs = sd.pair_list(V)
        cost = [nf[i] * nf[j] for i, j in pairs]
        load, want = [0] * world, [0] * len(pairs)
        for p in sorted(range(len(pairs)), key=lambda p: (-cost[p], p)):
            r = min(range(world), key=lambda k: (load[k], k))
            want[p] = r
            load[r] += cost[p]
        assert sd.assign_pairs(nf, world) == want, (nf, world)
    assert sd.assign_pairs([5], 4) == []
