"""generateMatchesExhaustive's merge on the device (csrc/merge.hip, ssrlcv_hip_merge_matches; src/MatchFactory.cu:943-1020):
the round-wise resolution of the seeds must return exactly what upstream's single-threaded walk returns (mode 1 of the
host test hook) -- on consistent scenes, on scenes where many seeds of an image share targets (the conflicts the rounds
exist for), and on chains of conflicts longer than the round limit (the in-order tail)."""
import numpy as np
import pytest
import torch

from test_merge_parallel import PAIR, random_pairs, run

pytestmark = pytest.mark.gpu


def device_merge(capi, nf, blocks):
    counts = [len(b) for b in blocks]
    allp = np.ascontiguousarray(np.concatenate(blocks)) if len(blocks) and sum(counts) else np.zeros(0, PAIR)
    pairs_d = capi.to_dev(allp) if len(allp) else torch.zeros(16, dtype=torch.uint8, device="cuda")
    mm_d, mem_d, n_mm, n_mem, rounds, _ = capi.merge_matches_device(nf, counts, pairs_d)
    mm = mm_d.cpu().numpy().view("<u4").reshape(-1, 2)[:n_mm]
    mem = mem_d.cpu().numpy().view("<u4").reshape(-1, 2)[:n_mem]
    return mm, mem, rounds


@pytest.mark.parametrize("V,n,density,spread", [(3, 2000, 0.5, 2000), (4, 3000, 0.6, 3000), (4, 5000, 0.7, 40),
                                                 (5, 1500, 0.9, 1500), (6, 800, 0.8, 25), (8, 400, 0.5, 400), (3, 50000, 0.4, 50000),
                                                 (4, 200000, 0.6, 200000), (12, 300, 0.7, 60), (32, 120, 0.4, 50)])
def test_device_merge_equals_the_sequential_walk(capi, V, n, density, spread):
    from ssrlcv_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(V * 1000 + n)
    for trial in range(3):
        nf = [int(n * (0.7 + 0.6 * rng.random())) for _ in range(V)]
        blocks = random_pairs(rng, nf, density, min(spread, min(nf)))
        mm1, mem1 = run(lib, nf, blocks, 1)
        mm0, mem0, rounds = device_merge(capi, nf, blocks)
        print("V=%d n=%d spread=%d: %d multi-matches, %d members, %d rounds" % (V, n, spread, len(mm1), len(mem1), rounds))
        assert len(mm1) > 0
        assert np.array_equal(mm0, mm1) and np.array_equal(mem0, mem1), (V, n, trial, len(mm0), len(mm1))


def test_a_chain_of_conflicts_longer_than_the_round_limit(capi):
    """Every seed of image 0 is matched to the SAME feature of image 1 (which goes on to image 2): each seed's outcome
    depends on the one before it, one seed resolves per round, and past the round limit the rest is walked in order."""
    from ssrlcv_amd import _lib
    lib = _lib.load()
    n = 300
    nf = [n, 8, 8]
    b01 = np.zeros(n, PAIR)
    b01["a"][:, 0], b01["a"][:, 1] = 0, np.arange(n)
    b01["b"][:, 0], b01["b"][:, 1] = 1, 3
    b02 = np.zeros(n // 2, PAIR)  # half of the seeds also see image 2 directly
    b02["a"][:, 0], b02["a"][:, 1] = 0, np.arange(0, n, 2)
    b02["b"][:, 0], b02["b"][:, 1] = 2, 5
    b12 = np.zeros(1, PAIR)
    b12["a"][0], b12["b"][0] = (1, 3), (2, 5)
    blocks = [b01, b02, b12]
    mm1, mem1 = run(lib, nf, blocks, 1)
    mm0, mem0, rounds = device_merge(capi, nf, blocks)
    print("%d multi-matches, %d rounds" % (len(mm1), rounds))
    assert rounds >= 40
    assert np.array_equal(mm0, mm1) and np.array_equal(mem0, mem1)


def test_empty_and_two_image_inputs(capi):
    from ssrlcv_amd import _lib
    lib = _lib.load()
    # two images: nothing seeds a multi-match (upstream only walks images 0..V-3)
    blk = np.zeros(3, PAIR)
    blk["a"][:, 1], blk["b"][:, 0], blk["b"][:, 1] = [0, 1, 2], 1, [2, 1, 0]
    mm0, mem0, _ = device_merge(capi, [4, 4], [blk])
    mm1, mem1 = run(lib, [4, 4], [blk], 1)
    assert len(mm0) == len(mm1) == 0 and len(mem0) == len(mem1) == 0
    # three images without a single match
    mm0, mem0, _ = device_merge(capi, [5, 5, 5], [np.zeros(0, PAIR)] * 3)
    assert len(mm0) == 0 and len(mem0) == 0


def test_invalid_entries_are_refused(capi):
    blk = np.zeros(1, PAIR)
    blk["a"][0], blk["b"][0] = (0, 5), (1, 999)  # b.y past image 1's feature array
    with pytest.raises(RuntimeError):
        device_merge(capi, [10, 10, 10], [blk, np.zeros(0, PAIR), np.zeros(0, PAIR)])
    dup = np.zeros(2, PAIR)  # one query matched twice in one pair: only the host walk takes it
    dup["a"][:, 1], dup["b"][:, 0], dup["b"][:, 1] = 4, 1, [2, 3]
    with pytest.raises(RuntimeError):
        device_merge(capi, [10, 10, 10], [dup, np.zeros(0, PAIR), np.zeros(0, PAIR)])


def test_match_set_of_the_flow_is_the_host_merge_s(capi, monkeypatch):
    """pipeline.build_match_set (device merge + device KeyPoint gather) against the same call on the host merge."""
    import helpers as H
    from ssrlcv_amd import pipeline
    rng = np.random.default_rng(11)
    nf = [4000, 3500, 4200, 3900]
    feats = []
    for n in nf:
        f = np.zeros(n, H.FEATURE)
        f["loc"] = rng.random((n, 2), dtype=np.float32) * 1000
        feats.append(capi.to_dev(f))
    blocks = random_pairs(rng, nf, 0.6, 3000)
    pair_tensors = [capi.to_dev(np.ascontiguousarray(b)) if len(b) else torch.zeros(0, dtype=torch.uint8, device="cuda") for b in blocks]
    dev = {}
    mm0, kp0 = pipeline.build_match_set(feats, pair_tensors, dev)
    monkeypatch.setenv("SSRLCV_MERGE_HOST", "1")
    mm1, kp1 = pipeline.build_match_set(feats, pair_tensors, {})
    assert len(mm0) > 500 and np.array_equal(mm0, mm1) and np.array_equal(kp0, kp1)
    assert dev["matches"].numel() == 8 * len(mm0) and dev["keypoints"].numel() >= 16 * len(kp0)
