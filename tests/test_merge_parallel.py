"""generateMatchesExhaustive's host merge (src/MatchFactory.cu:943-1020): the parallel version the library runs by default
must return exactly what upstream's single-threaded walk returns (mode 1 of the test hook), also when many seeds of an
image share lists of later images -- the case its conflict detection exists for.  No GPU needed."""
import ctypes

import numpy as np
import pytest

from ssrlcv_amd import _lib

PAIR = np.dtype([("a", "<u4", (2,)), ("b", "<u4", (2,))])


def run(lib, num_features, blocks, mode):
    counts = np.array([len(b) for b in blocks], np.uint32)
    allp = np.ascontiguousarray(np.concatenate(blocks)) if len(blocks) and counts.sum() else np.zeros(0, PAIR)
    nf = np.array(num_features, np.uint32)
    mm_p, mem_p = ctypes.c_void_p(), ctypes.c_void_p()
    nmm, nmem = ctypes.c_uint32(), ctypes.c_uint32()
    rc = lib.ssrlcv_merge_matches_host_mode(ctypes.c_uint32(len(nf)), nf.ctypes.data_as(ctypes.c_void_p),
                                            ctypes.c_uint32(len(counts)), counts.ctypes.data_as(ctypes.c_void_p),
                                            allp.ctypes.data_as(ctypes.c_void_p), ctypes.byref(mm_p), ctypes.byref(mem_p),
                                            ctypes.byref(nmm), ctypes.byref(nmem), ctypes.c_int(mode))
    assert rc == 0, rc
    mm = np.ctypeslib.as_array(ctypes.cast(mm_p, ctypes.POINTER(ctypes.c_uint32)), shape=(max(nmm.value, 1), 2))[: nmm.value].copy()
    mem = np.ctypeslib.as_array(ctypes.cast(mem_p, ctypes.POINTER(ctypes.c_uint32)), shape=(max(nmem.value, 1), 2))[: nmem.value].copy()
    lib.ssrlcv_host_free(mm_p)
    lib.ssrlcv_host_free(mem_p)
    return mm, mem


def random_pairs(rng, num_features, density, spread):
    """Validated pair blocks in the reference's pair order; `spread` small = many queries hit the same target."""
    V = len(num_features)
    blocks = []
    for q in range(V - 1):
        for t in range(q + 1, V):
            keep = np.nonzero(rng.random(num_features[q]) < density)[0]
            blk = np.zeros(len(keep), PAIR)
            blk["a"][:, 0], blk["a"][:, 1] = q, keep
            blk["b"][:, 0] = t
            # a noisy monotone map (a consistent scene) folded into `spread` targets
            tgt = (keep * num_features[t] // max(num_features[q], 1) + rng.integers(-2, 3, len(keep))) % spread
            blk["b"][:, 1] = np.clip(tgt, 0, num_features[t] - 1)
            blocks.append(blk)
    return blocks


@pytest.mark.parametrize("V,n,density,spread", [(3, 2000, 0.5, 2000), (4, 3000, 0.6, 3000), (4, 5000, 0.7, 40),
                                                 (5, 1500, 0.9, 1500), (6, 800, 0.8, 25), (8, 400, 0.5, 400), (3, 50000, 0.4, 50000)])
def test_parallel_merge_equals_the_sequential_walk(V, n, density, spread):
    lib = _lib.load()
    rng = np.random.default_rng(V * 1000 + n)
    for trial in range(3):
        nf = [int(n * (0.7 + 0.6 * rng.random())) for _ in range(V)]
        blocks = random_pairs(rng, nf, density, min(spread, min(nf)))
        mm1, mem1 = run(lib, nf, blocks, 1)
        mm0, mem0 = run(lib, nf, blocks, 0)
        assert len(mm1) > 0
        assert np.array_equal(mm0, mm1) and np.array_equal(mem0, mem1), (V, n, trial, len(mm0), len(mm1))


def test_invalid_entries_are_refused():
    lib = _lib.load()
    blk = np.zeros(1, PAIR)
    blk["a"][0], blk["b"][0] = (0, 5), (1, 999)  # b.y past image 1's feature array
    counts = np.array([1, 0, 0], np.uint32)
    nf = np.array([10, 10, 10], np.uint32)
    mm_p, mem_p = ctypes.c_void_p(), ctypes.c_void_p()
    nmm, nmem = ctypes.c_uint32(), ctypes.c_uint32()
    rc = lib.ssrlcv_merge_matches_host(ctypes.c_uint32(3), nf.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(3),
                                       counts.ctypes.data_as(ctypes.c_void_p), blk.ctypes.data_as(ctypes.c_void_p),
                                       ctypes.byref(mm_p), ctypes.byref(mem_p), ctypes.byref(nmm), ctypes.byref(nmem))
    assert rc == -1
