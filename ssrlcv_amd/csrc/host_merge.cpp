// ssrlcv_amd/csrc/host_merge.cpp -- host half of MatchFactory<T>::generateMatchesExhaustive
// (src/MatchFactory.cu:943-1020): adjacency lists per (image, feature) in pair order, transitive-consistency walk
// with std::set_intersection, flattening to MultiMatch{n,index} + member list.
//
// Upstream is one thread walking the seed features in order; the walk of a seed reads and clears lists of later
// images, so the order is part of the result.  This version gives the same MatchSet (tests/test_merge_parallel.py holds
// it to the literal sequential walk, which stays here as the checker's reference and as the path for conflicting seeds)
// with the work spread over the host cores, because on a multi-GPU run the merge is replicated on every rank and was a
// third of the 4 x 4096^2 step while SIFT and matching shrink with the rank count:
//   * the adjacency lists are one CSR array, counted and filled in parallel (a list receives at most one entry per
//     image pair, so the pair blocks are filled one after the other and the entries of a block land independently);
//   * seeds are processed image by image.  Within an image the walks only touch lists of LATER images; a seed whose
//     potential reads and writes there are shared with no other seed of its image ("clean") gives the same outcome in any
//     order and is resolved in parallel from the state at the start of the image; the few seeds that share a list with
//     another one are then walked sequentially, in feature order, on the live state -- exactly upstream's order among
//     themselves, and by construction independent of the clean ones;
//   * the MultiMatch / member arrays are stitched in (image, feature) order from per-seed results by a prefix sum.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <mutex>
#include <omp.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include "dev_switch.h"
#include "ssrlcv_hip.h"

namespace {
struct U2 {
  uint32_t x, y;
};
inline bool operator==(const U2& a, const U2& b) { return a.x == b.x && a.y == b.y; }
inline bool operator<(const U2& a, const U2& b) {  // src/cuda_vec_util.cu:559-563
  if (a == b) return false;
  else if (a.x == b.x) return a.y < b.y;
  else return a.x < b.x;
}

struct Lists {
  uint32_t V;
  std::vector<size_t> base;     // first list slot of image i (images 0..V-2 own lists)
  std::vector<uint32_t> start;  // CSR offsets
  std::vector<uint32_t> len;    // mutable list sizes (clear() = 0)
  std::vector<U2> entries;
  size_t list_of(uint32_t img, uint32_t feat) const { return base[img] + feat; }
};

enum Outcome { kSkip = 0, kGood = 1, kBad = 2 };

// The consistency walk of one seed list `a` (src/MatchFactory.cu:971-1004) on the current lengths.  onRead(l) is called
// for every list of a later image whose length or entries the walk looks at.
template <typename ReadFn>
inline Outcome walk(const Lists& L, size_t a, ReadFn onRead) {
  if (L.len[a] == 0) return kSkip;
  const uint32_t V = L.V;
  size_t prev = a;
  while (true) {
    const U2 head = L.entries[L.start[prev]];
    if (head.x == V - 1) break;
    const size_t next = L.list_of(head.x, head.y);
    onRead(next);
    if (L.len[next] == 0) break;
    // |set_intersection(list[prev], list[next])| (:984-990) without materialising it: both lists are sorted by the
    // same operator< (an image's list holds at most one entry per later image, appended in pair order), and
    // std::set_intersection's merge is replayed literally
    uint32_t common = 0;
    {
      const U2* p1 = &L.entries[L.start[prev]];
      const U2* e1 = p1 + L.len[prev];
      const U2* p2 = &L.entries[L.start[next]];
      const U2* e2 = p2 + L.len[next];
      while (p1 != e1 && p2 != e2) {
        if (*p1 < *p2) ++p1;
        else if (*p2 < *p1) ++p2;
        else { ++common; ++p1; ++p2; }
      }
    }
    if (common != L.len[next]) return kBad;
    else if (L.len[next] == 1) break;
    else prev = next;
  }
  return kGood;
}
// the lists a good seed clears (:1012-1016: every member but the last, stopping at the last image)
template <typename WriteFn>
inline void for_each_cleared(const Lists& L, size_t a, WriteFn onWrite) {
  for (uint32_t m = 0; m + 1 < L.len[a]; ++m) {
    const U2 e = L.entries[L.start[a] + m];
    if (e.x == L.V - 1) break;
    onWrite(L.list_of(e.x, e.y));
  }
}
inline void commit(Lists& L, size_t a, Outcome o) {
  if (o == kBad) L.len[a] = 0;
  else if (o == kGood) for_each_cleared(L, a, [&](size_t l) { L.len[l] = 0; });
}

int build_lists(Lists& L, const uint32_t* numFeatures, uint32_t numPairs, const uint32_t* pairCounts, const ssrlcv_uint2_pair* pairs) {
  const uint32_t V = L.V;
  L.base.assign(V, 0);
  for (uint32_t i = 0; i + 1 < V; ++i) L.base[i + 1] = L.base[i] + numFeatures[i];
  const size_t numLists = L.base[V - 1];
  std::vector<size_t> blockStart(numPairs + 1, 0);
  for (uint32_t k = 0; k < numPairs; ++k) blockStart[k + 1] = blockStart[k] + pairCounts[k];
  const size_t total = blockStart[numPairs];
  if (total > 0xffffffffull) return SSRLCV_ERR_CAPACITY;
  L.start.assign(numLists + 1, 0);
  L.len.assign(numLists, 0);
  // a names a list owner (images 0..V-2), b a feature of a later image: anything else (an un-compacted "invalid" a == b
  // entry of the matcher, an index past a feature array) would index the lists out of bounds
  int bad = 0;
  uint32_t* cnt = L.start.data() + 1;
#pragma omp parallel for schedule(static) reduction(| : bad)
  for (long long e = 0; e < (long long)total; ++e) {
    const ssrlcv_uint2_pair& p = pairs[e];
    if (p.a.x >= V - 1 || p.a.y >= numFeatures[p.a.x] || p.b.x >= V || p.b.y >= numFeatures[p.b.x] || p.b.x <= p.a.x) {
      bad |= 1;
      continue;
    }
    __atomic_fetch_add(&cnt[L.base[p.a.x] + p.a.y], 1u, __ATOMIC_RELAXED);
  }
  if (bad) return SSRLCV_ERR_INVALID_ARG;
  for (size_t l = 0; l < numLists; ++l) L.start[l + 1] += L.start[l];
  L.entries.resize(total ? total : 1);
  // Entries are appended in pair order exactly like upstream's push_back.  A validated pair block holds a query at most
  // once, so within a block every entry goes to a different list: the blocks run one after the other, their entries in
  // parallel (the fetch-add keeps a block with a repeated query well defined up to the order of the repeats).
  for (uint32_t k = 0; k < numPairs; ++k) {
    const ssrlcv_uint2_pair* blk = pairs + blockStart[k];
    const long long n = (long long)pairCounts[k];
#pragma omp parallel for schedule(static) if (n > 4096)
    for (long long e = 0; e < n; ++e) {
      const size_t l = L.base[blk[e].a.x] + blk[e].a.y;
      const uint32_t slot = __atomic_fetch_add(&L.len[l], 1u, __ATOMIC_RELAXED);
      L.entries[L.start[l] + slot] = U2{blk[e].b.x, blk[e].b.y};
    }
  }
  return SSRLCV_OK;
}

void emit(const Lists& L, uint32_t i, uint32_t f, size_t a, uint32_t memIndex, ssrlcv_multimatch* mm, U2* mem) {
  mm->numKeyPoints = L.len[a] + 1;
  mm->index = (int)memIndex;
  mem[0] = U2{i, f};
  std::memcpy(mem + 1, &L.entries[L.start[a]], sizeof(U2) * L.len[a]);
}

}  // namespace

extern "C" {

// mode: 0 = parallel, 1 = the literal sequential walk.  ssrlcv_merge_matches_host picks one (below).
int ssrlcv_merge_matches_host_mode(uint32_t numImages, const uint32_t* numFeatures, uint32_t numPairs, const uint32_t* pairCounts,
                                   const ssrlcv_uint2_pair* pairs, ssrlcv_multimatch** matches_out, ssrlcv_uint2** members_out,
                                   uint32_t* numMatches, uint32_t* numMembers, int mode) {
  if (numImages < 2 || !numFeatures || (numPairs && (!pairCounts || !pairs)) || !matches_out || !members_out || !numMatches ||
      !numMembers)
    return SSRLCV_ERR_INVALID_ARG;
  // The working arrays (tens of MB for four 4096^2 views) live in one grow-only arena kept between calls: freshly mapped
  // vectors cost more in page faults than the merge itself takes.  One merge at a time per process (the lock).
  static std::mutex arenaLock;
  std::lock_guard<std::mutex> guard(arenaLock);
  static Lists L;
  static std::vector<uint32_t> readers, writers, mmOff, memOff;
  static std::vector<uint8_t> outcome, dirty;
  static std::vector<ssrlcv_multimatch> mm;
  static std::vector<U2> mem;
  mm.clear();
  mem.clear();
  L.V = numImages;
  const uint32_t V = numImages;
  static const bool timing = svdev::env("SSRLCV_MERGE_TIMING") != nullptr;
  auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "merge %-10s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  };
  int rc = build_lists(L, numFeatures, numPairs, pairCounts, pairs);
  if (rc) return rc;
  lap("lists");
  if (mode == 1) {
    mm.reserve(L.entries.size() / 2 + 16);
    mem.reserve(L.entries.size() + L.entries.size() / 2 + 16);
    for (uint32_t i = 0; i + 2 < V; ++i) {  // only images 0..V-3 seed multi-matches (:969)
      const uint32_t nf = numFeatures[i];
      for (uint32_t f = 0; f < nf; ++f) {
        const size_t a = L.list_of(i, f);
        const Outcome o = walk(L, a, [](size_t) {});
        if (o == kGood) {
          mm.emplace_back();
          const size_t at = mem.size();
          mem.resize(at + L.len[a] + 1);
          emit(L, i, f, a, (uint32_t)at, &mm.back(), &mem[at]);
        }
        commit(L, a, o);
      }
    }
  } else {
    const size_t numLists = L.len.size();
    uint32_t maxSeeds = 0;
    size_t seedTotal = 0;
    for (uint32_t i = 0; i + 2 < V; ++i) {
      maxSeeds = numFeatures[i] > maxSeeds ? numFeatures[i] : maxSeeds;
      seedTotal += numFeatures[i];
    }
    if (readers.size() < numLists) { readers.resize(numLists); writers.resize(numLists); }
    if (outcome.size() < maxSeeds) { outcome.resize(maxSeeds); dirty.resize(maxSeeds); }
    if (mmOff.size() < (size_t)maxSeeds + 1) { mmOff.resize((size_t)maxSeeds + 1); memOff.resize((size_t)maxSeeds + 1); }
    mm.reserve(seedTotal < L.entries.size() ? seedTotal : L.entries.size());
    mem.reserve(L.entries.size() + mm.capacity());
    for (uint32_t i = 0; i + 2 < V; ++i) {
      const long long nf = (long long)numFeatures[i];
      // marks of lists of later images only, reset for this image's phase
      const long long lo = (long long)L.base[i + 1], hi = (long long)numLists;
#pragma omp parallel for schedule(static)
      for (long long l = lo; l < hi; ++l) readers[l] = writers[l] = 0u;
#pragma omp parallel for schedule(static)
      for (long long f = 0; f < nf; ++f) { outcome[f] = kSkip; dirty[f] = 0; }
      // 1. outcome on the state at the start of the image + marks of everything the seed may read or clear.  (On the
      //    live state a walk can only stop earlier -- lengths only ever drop to 0 -- so these are supersets; the clears
      //    are marked whatever the outcome, because a seed that is bad now may be good at its turn.)
#pragma omp parallel for schedule(dynamic, 2048)
      for (long long f = 0; f < nf; ++f) {
        const size_t a = L.list_of(i, (uint32_t)f);
        if (L.len[a] == 0) continue;
        outcome[f] = (uint8_t)walk(L, a, [&](size_t l) { __atomic_fetch_add(&readers[l], 1u, __ATOMIC_RELAXED); });
        for_each_cleared(L, a, [&](size_t l) { __atomic_fetch_add(&writers[l], 1u, __ATOMIC_RELAXED); });
      }
      // 2. a seed is clean when nobody else clears what it reads or clears, and nobody else reads what it clears
      //    (a list is marked once per walk step / member, so its own marks are counted by walking again)
#pragma omp parallel for schedule(dynamic, 2048)
      for (long long f = 0; f < nf; ++f) {
        if (outcome[f] == kSkip) continue;
        const size_t a = L.list_of(i, (uint32_t)f);
        size_t mineR[64], mineW[64];
        uint32_t nr = 0, nw = 0;
        bool overflow = false;
        walk(L, a, [&](size_t l) { if (nr < 64) mineR[nr++] = l; else overflow = true; });
        for_each_cleared(L, a, [&](size_t l) { if (nw < 64) mineW[nw++] = l; else overflow = true; });
        bool d = overflow;
        auto own = [](const size_t* v, uint32_t n, size_t l) { uint32_t c = 0; for (uint32_t k = 0; k < n; ++k) c += v[k] == l; return c; };
        for (uint32_t k = 0; k < nr && !d; ++k) d = writers[mineR[k]] > own(mineW, nw, mineR[k]);
        for (uint32_t k = 0; k < nw && !d; ++k)
          d = writers[mineW[k]] > own(mineW, nw, mineW[k]) || readers[mineW[k]] > own(mineR, nr, mineW[k]);
        dirty[f] = d;
      }
      // 3. clean seeds commit in parallel (their lists are theirs alone) ...
#pragma omp parallel for schedule(dynamic, 2048)
      for (long long f = 0; f < nf; ++f)
        if (outcome[f] != kSkip && !dirty[f]) commit(L, L.list_of(i, (uint32_t)f), (Outcome)outcome[f]);
      // 4. ... the others in upstream's order on the live state.  A good seed's own list is final when it commits (only
      //    seeds of EARLIER images clear lists of this image), so its members are emitted afterwards from L.
      for (long long f = 0; f < nf; ++f) {
        if (!dirty[f]) continue;
        const size_t a = L.list_of(i, (uint32_t)f);
        const Outcome o = walk(L, a, [](size_t) {});
        outcome[f] = (uint8_t)o;
        commit(L, a, o);
      }
      if (timing) {
        long long nd = 0, ns = 0;
        for (long long f = 0; f < nf; ++f) { nd += dirty[f]; ns += outcome[f] != kSkip; }
        std::fprintf(stderr, "merge image %u: %lld seeds with a list, %lld of them share a list with another seed\n", i, ns, nd);
      }
      // 5. stitch in feature order
      mmOff[0] = memOff[0] = 0;
      for (long long f = 0; f < nf; ++f) {
        const bool good = outcome[f] == kGood;
        mmOff[f + 1] = mmOff[f] + (good ? 1u : 0u);
        memOff[f + 1] = memOff[f] + (good ? L.len[L.list_of(i, (uint32_t)f)] + 1u : 0u);
      }
      const size_t mm0 = mm.size(), mem0 = mem.size();
      if (mem0 + memOff[nf] > 0x7fffffffull) return SSRLCV_ERR_CAPACITY;  // MultiMatch::index is an int
      mm.resize(mm0 + mmOff[nf]);
      mem.resize(mem0 + memOff[nf]);
#pragma omp parallel for schedule(static)
      for (long long f = 0; f < nf; ++f)
        if (outcome[f] == kGood)
          emit(L, i, (uint32_t)f, L.list_of(i, (uint32_t)f), (uint32_t)(mem0 + memOff[f]), &mm[mm0 + mmOff[f]], &mem[mem0 + memOff[f]]);
    }
  }
  lap("walk");
  *numMatches = (uint32_t)mm.size();
  *numMembers = (uint32_t)mem.size();
  *matches_out = (ssrlcv_multimatch*)std::malloc(sizeof(ssrlcv_multimatch) * (mm.size() ? mm.size() : 1));
  *members_out = (ssrlcv_uint2*)std::malloc(sizeof(ssrlcv_uint2) * (mem.size() ? mem.size() : 1));
  if (!*matches_out || !*members_out) return SSRLCV_ERR_INVALID_ARG;
  if (!mm.empty()) std::memcpy(*matches_out, mm.data(), sizeof(ssrlcv_multimatch) * mm.size());
  if (!mem.empty()) std::memcpy(*members_out, mem.data(), sizeof(ssrlcv_uint2) * mem.size());
  return SSRLCV_OK;
}

int ssrlcv_merge_matches_host(uint32_t numImages, const uint32_t* numFeatures, uint32_t numPairs, const uint32_t* pairCounts,
                              const ssrlcv_uint2_pair* pairs, ssrlcv_multimatch** matches_out, ssrlcv_uint2** members_out,
                              uint32_t* numMatches, uint32_t* numMembers) {
  // The parallel walk does three passes over the seeds instead of one, so it pays from about six threads up.  Measured on
  // the GPU box (4 x 4096^2 views, 0.63 M pairs): C merge 12 ms with a team of 8 against 20 ms sequential -- but the
  // whole merge stage of the flow took 29 ms against 21: the box grants the process a CPU quota (cgroup), the OpenMP
  // burst spends it, and the main thread is then throttled through the copies that follow (single H2D copies of 2 MB
  // measured at 9-22 ms right after it; with a team of 16 the merge itself jittered between 10 and 180 ms).  So the
  // default is the sequential walk; SSRLCV_MERGE_THREADS=<n> enables the parallel one where the cores are really there.
  static const int threads = [] {
    if (const char* e = svdev::env("SSRLCV_MERGE_THREADS")) return std::atoi(e) > 0 ? std::atoi(e) : 1;
    return 1;
  }();
  if (threads <= 1)
    return ssrlcv_merge_matches_host_mode(numImages, numFeatures, numPairs, pairCounts, pairs, matches_out, members_out,
                                          numMatches, numMembers, 1);
  const int before = omp_get_max_threads();
  omp_set_num_threads(threads);
  const int rc = ssrlcv_merge_matches_host_mode(numImages, numFeatures, numPairs, pairCounts, pairs, matches_out, members_out,
                                                numMatches, numMembers, 0);
  omp_set_num_threads(before);
  return rc;
}

void ssrlcv_host_free(void* p) { std::free(p); }

// The pair table of the sharded generateMatchesExhaustive: which rank matches image pair p (pairs
// in upstream's order (0,1),(0,2)..(1,2).., src/MatchFactory.cu:924-936).  The cost of a pair is nq x nt distance
// evaluations, known on every rank once the feature counts are exchanged; longest-processing-time-first -- pairs by
// descending cost (ties by pair index) each go to the least loaded rank (ties to the lowest rank) -- is deterministic, so
// every rank derives the same table.  ONE definition for the Python driver (ssrlcv_amd/dist.py) and the C++ one
// (host/Distributed.hpp).
int ssrlcv_assign_pairs_host(uint32_t numImages, const uint32_t* numFeatures, uint32_t world, uint32_t* owners_out) {
  if (numImages < 2 || !numFeatures || world == 0 || !owners_out) return SSRLCV_ERR_INVALID_ARG;
  std::vector<unsigned long long> cost;
  for (uint32_t i = 0; i + 1 < numImages; ++i)
    for (uint32_t j = i + 1; j < numImages; ++j) cost.push_back((unsigned long long)numFeatures[i] * numFeatures[j]);
  std::vector<uint32_t> order(cost.size());
  for (uint32_t p = 0; p < order.size(); ++p) order[p] = p;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
  std::vector<unsigned long long> load(world, 0ull);
  for (uint32_t p : order) {
    uint32_t r = 0;
    for (uint32_t k = 1; k < world; ++k)
      if (load[k] < load[r]) r = k;
    owners_out[p] = r;
    load[r] += cost[p];
  }
  return SSRLCV_OK;
}

}  // extern "C"
