// ssrlcv_amd/csrc/merge.hip -- device form of the host half of MatchFactory<T>::generateMatchesExhaustive
// (src/MatchFactory.cu:943-1020): adjacency lists per (image, feature), the transitive-consistency walk of every seed
// feature, MultiMatch{n, index} + member list in (image, feature) order.  Same MatchSet as upstream's single-threaded walk
// (ssrlcv_merge_matches_host_mode(..., 1), held to it by tests/test_gpu_merge.py), without the D2H copy of the matches,
// the host's pointer chasing and the H2D copy of the members that made the merge a quarter of the 4 x 4096^2 step.
//
// Upstream walks the seeds in order; the walk of a seed reads lists of LATER images and, when the seed is accepted,
// clears the lists of its members, so the order is part of the result.  Here the seeds of an image are resolved in
// ROUNDS.  R(f) = the lists f's walk reads on the current state, W(f) = the lists f clears if it is accepted (the members
// of its own list: fixed).  Lengths only ever drop to zero and a walk only stops earlier on shorter lists, so R(f) taken
// now is a superset of what f reads at its turn.  A seed is READY when no unresolved seed g < f has W(g) meeting R(f)
// (g would change what f sees) or R(g) meeting W(f) (f would change what g sees); the lowest unresolved seed always is.
// Ready seeds are independent of each other and of every earlier unresolved seed, so a round resolves them all at once
// from the state at its start and the result is upstream's.  Conflicts are rare and short (four 4096^2 views: 3 % of the
// seeds share a list with another seed, chains of two or three): a handful of rounds; after kMaxRounds the remaining seeds
// are walked by one thread in feature order.
//   lists   : a list gets at most one entry per partner image (the matcher returns one match per query and pair), and
//             upstream appends in pair order = ascending partner image: the slot of an entry is the number of smaller
//             partner images present, from a per-list bit mask -- no ordering between threads needed.  (Two entries of one
//             list with the same partner image, which upstream would append twice, are refused: SSRLCV_ERR_INVALID_ARG.)
//   output  : exclusive scans over the seeds' (accepted ? 1 : 0) and (accepted ? members : 0).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>  // rocprim's texture_cache_iterator.hpp calls memset without including it
#include <rocprim/rocprim.hpp>
#include "device_math.h"
#include "ssrlcv_hip.h"

namespace {

constexpr int kMaxImages = 32;   // partner-image masks are 32-bit
constexpr int kMaxRounds = 48;

struct U2 {
  uint32_t x, y;
};
__device__ __forceinline__ bool u2_eq(const U2& a, const U2& b) { return a.x == b.x && a.y == b.y; }
__device__ __forceinline__ bool u2_lt(const U2& a, const U2& b) {  // src/cuda_vec_util.cu:559-563
  if (u2_eq(a, b)) return false;
  else if (a.x == b.x) return a.y < b.y;
  else return a.x < b.x;
}

struct Lists {
  uint32_t V;
  uint32_t base[kMaxImages + 1];  // first list of image i (images 0..V-2 own lists); base[V-1] = number of lists
  uint32_t nf[kMaxImages];
  const uint32_t* start;          // CSR offsets
  uint32_t* len;                  // live lengths (clear = 0)
  const U2* entries;
  __device__ __forceinline__ uint32_t list_of(uint32_t img, uint32_t feat) const { return base[img] + feat; }
};

enum Outcome : uint8_t { kSkip = 0, kGood = 1, kBad = 2 };

// the consistency walk of one seed list (src/MatchFactory.cu:971-1004) on the live lengths; onRead(l) for every list of a
// later image whose length or entries the walk looks at
template <typename ReadFn>
__device__ __forceinline__ Outcome walk(const Lists& L, uint32_t a, ReadFn onRead) {
  if (L.len[a] == 0) return kSkip;
  const uint32_t V = L.V;
  uint32_t prev = a;
  while (true) {
    const U2 head = L.entries[L.start[prev]];
    if (head.x == V - 1) break;
    const uint32_t next = L.list_of(head.x, head.y);
    onRead(next);
    const uint32_t lenNext = L.len[next];
    if (lenNext == 0) break;
    // |set_intersection(list[prev], list[next])| (:984-990): both lists are sorted by the same operator<
    uint32_t common = 0;
    {
      const U2* p1 = L.entries + L.start[prev];
      const U2* e1 = p1 + L.len[prev];
      const U2* p2 = L.entries + L.start[next];
      const U2* e2 = p2 + lenNext;
      while (p1 != e1 && p2 != e2) {
        const U2 v1 = *p1, v2 = *p2;
        if (u2_lt(v1, v2)) ++p1;
        else if (u2_lt(v2, v1)) ++p2;
        else { ++common; ++p1; ++p2; }
      }
    }
    if (common != lenNext) return kBad;
    else if (lenNext == 1) break;
    else prev = next;
  }
  return kGood;
}
// the lists an accepted seed clears (:1012-1016: every member but the last, stopping at the last image)
template <typename WriteFn>
__device__ __forceinline__ void for_each_cleared(const Lists& L, uint32_t a, WriteFn onWrite) {
  const uint32_t n = L.len[a];
  for (uint32_t m = 0; m + 1 < n; ++m) {
    const U2 e = L.entries[L.start[a] + m];
    if (e.x == L.V - 1) break;
    onWrite(L.list_of(e.x, e.y));
  }
}
__device__ __forceinline__ void commit(const Lists& L, uint32_t a, Outcome o) {
  if (o == kBad) L.len[a] = 0;
  else if (o == kGood) for_each_cleared(L, a, [&](uint32_t l) { L.len[l] = 0; });
}

// ---- lists
__global__ __launch_bounds__(256) void k_merge_masks(Lists L, const ssrlcv_uint2_pair* __restrict__ pairs, uint32_t total,
                                                     uint32_t* __restrict__ mask, int* __restrict__ bad) {
  const uint32_t e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const ssrlcv_uint2_pair p = pairs[e];
  const uint32_t V = L.V;
  if (p.a.x >= V - 1 || p.b.x >= V || p.b.x <= p.a.x || p.a.y >= L.nf[p.a.x] || p.b.y >= L.nf[p.b.x]) {
    atomicOr(bad, 1);
    return;
  }
  const uint32_t old = atomicOr(&mask[L.list_of(p.a.x, p.a.y)], 1u << p.b.x);
  if (old & (1u << p.b.x)) atomicOr(bad, 2);  // a second entry with the same partner image
}
__global__ __launch_bounds__(256) void k_merge_len(const uint32_t* __restrict__ mask, uint32_t numLists, uint32_t* __restrict__ len) {
  const uint32_t l = blockIdx.x * 256 + threadIdx.x;
  if (l < numLists) len[l] = (uint32_t)__popc(mask[l]);
}
__global__ __launch_bounds__(256) void k_merge_fill(Lists L, const ssrlcv_uint2_pair* __restrict__ pairs, uint32_t total,
                                                    const uint32_t* __restrict__ mask, U2* __restrict__ entries) {
  const uint32_t e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const ssrlcv_uint2_pair p = pairs[e];
  const uint32_t l = L.list_of(p.a.x, p.a.y);
  const uint32_t slot = (uint32_t)__popc(mask[l] & ((1u << p.b.x) - 1u));
  entries[L.start[l] + slot] = U2{p.b.x, p.b.y};
}

// ---- rounds of image i.  state[f]: 1 = unresolved
__global__ __launch_bounds__(256) void k_merge_begin(Lists L, uint32_t i, uint8_t* __restrict__ state, uint8_t* __restrict__ outcome) {
  const uint32_t f = blockIdx.x * 256 + threadIdx.x;
  if (f >= L.nf[i]) return;
  state[f] = L.len[L.list_of(i, f)] != 0;
  outcome[f] = kSkip;
}
__global__ __launch_bounds__(256) void k_merge_mark(Lists L, uint32_t i, const uint8_t* __restrict__ state, uint32_t* __restrict__ minReader,
                                                    uint32_t* __restrict__ minWriter, uint32_t* __restrict__ unresolved) {
  const uint32_t f = blockIdx.x * 256 + threadIdx.x;
  const bool live = f < L.nf[i] && state[f];
  if (live) {
    const uint32_t a = L.list_of(i, f);
    walk(L, a, [&](uint32_t l) { atomicMin(&minReader[l], f); });
    for_each_cleared(L, a, [&](uint32_t l) { atomicMin(&minWriter[l], f); });
  }
  const unsigned long long b = __ballot(live);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(unresolved, (uint32_t)__popcll(b));
}
// ready: no unresolved lower seed clears what f reads, none reads what f clears (the marks of f itself are f)
__global__ __launch_bounds__(256) void k_merge_ready(Lists L, uint32_t i, uint8_t* __restrict__ state, const uint32_t* __restrict__ minReader,
                                                     const uint32_t* __restrict__ minWriter) {
  const uint32_t f = blockIdx.x * 256 + threadIdx.x;
  if (f >= L.nf[i] || !state[f]) return;
  const uint32_t a = L.list_of(i, f);
  bool ready = true;
  walk(L, a, [&](uint32_t l) { ready = ready && minWriter[l] >= f; });
  for_each_cleared(L, a, [&](uint32_t l) { ready = ready && minReader[l] >= f; });
  if (ready) state[f] = 2;
}
__global__ __launch_bounds__(256) void k_merge_commit(Lists L, uint32_t i, uint8_t* __restrict__ state, uint8_t* __restrict__ outcome) {
  const uint32_t f = blockIdx.x * 256 + threadIdx.x;
  if (f >= L.nf[i] || state[f] != 2) return;
  const uint32_t a = L.list_of(i, f);
  const Outcome o = walk(L, a, [](uint32_t) {});
  outcome[f] = o;
  commit(L, a, o);
  state[f] = 0;
}
// what is left after kMaxRounds, in upstream's order
__global__ void k_merge_sequential(Lists L, uint32_t i, uint8_t* __restrict__ state, uint8_t* __restrict__ outcome) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  for (uint32_t f = 0; f < L.nf[i]; ++f) {
    if (!state[f]) continue;
    const uint32_t a = L.list_of(i, f);
    const Outcome o = walk(L, a, [](uint32_t) {});
    outcome[f] = o;
    commit(L, a, o);
    state[f] = 0;
  }
}
__global__ __launch_bounds__(256) void k_merge_tally(Lists L, uint32_t i, const uint8_t* __restrict__ outcome, uint32_t seedBase,
                                                     uint32_t* __restrict__ good, uint32_t* __restrict__ members) {
  const uint32_t f = blockIdx.x * 256 + threadIdx.x;
  if (f >= L.nf[i]) return;
  const bool g = outcome[f] == kGood;
  good[seedBase + f] = g ? 1u : 0u;
  members[seedBase + f] = g ? L.len[L.list_of(i, f)] + 1u : 0u;  // an accepted seed's own list is final: only seeds of earlier images clear it
}
__global__ __launch_bounds__(256) void k_merge_emit(Lists L, uint32_t i, uint32_t seedBase, const uint32_t* __restrict__ good,
                                                    const uint32_t* __restrict__ mmOff, const uint32_t* __restrict__ memOff,
                                                    ssrlcv_multimatch* __restrict__ mm, ssrlcv_uint2* __restrict__ mem) {
  const uint32_t f = blockIdx.x * 256 + threadIdx.x;
  if (f >= L.nf[i] || !good[seedBase + f]) return;
  const uint32_t a = L.list_of(i, f), n = L.len[a], at = memOff[seedBase + f];
  ssrlcv_multimatch m;
  m.numKeyPoints = n + 1;
  m.index = (int)at;
  mm[mmOff[seedBase + f]] = m;
  mem[at].x = i;
  mem[at].y = f;
  for (uint32_t k = 0; k < n; ++k) {
    const U2 e = L.entries[L.start[a] + k];
    mem[at + 1 + k].x = e.x;
    mem[at + 1 + k].y = e.y;
  }
}
__global__ void k_merge_totals(const uint32_t* __restrict__ good, const uint32_t* __restrict__ members, const uint32_t* __restrict__ mmOff,
                               const uint32_t* __restrict__ memOff, uint32_t numSeeds, uint32_t* __restrict__ counts) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    counts[0] = numSeeds ? mmOff[numSeeds - 1] + good[numSeeds - 1] : 0u;
    counts[1] = numSeeds ? memOff[numSeeds - 1] + members[numSeeds - 1] : 0u;
  }
}

inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }
struct Layout {
  size_t mask, len, start, entries, minReader, minWriter, state, outcome, good, members, mmOff, memOff, scalars, scanTmp, total;
  size_t scanTmpBytes;
};
Layout make_layout(uint32_t numLists, uint32_t total, uint32_t maxSeeds, uint32_t numSeeds) {
  Layout y;
  size_t p = 0;
  auto take = [&](size_t bytes) { const size_t at = p; p += up256(bytes); return at; };
  y.mask = take((size_t)numLists * 4);
  y.len = take((size_t)numLists * 4);
  y.start = take(((size_t)numLists + 1) * 4);
  y.entries = take((size_t)(total ? total : 1) * 8);
  y.minReader = take((size_t)numLists * 4);
  y.minWriter = take((size_t)numLists * 4);
  y.state = take(maxSeeds ? maxSeeds : 1);
  y.outcome = take(maxSeeds ? maxSeeds : 1);
  y.good = take((size_t)(numSeeds ? numSeeds : 1) * 4);
  y.members = take((size_t)(numSeeds ? numSeeds : 1) * 4);
  y.mmOff = take((size_t)(numSeeds ? numSeeds : 1) * 4);
  y.memOff = take((size_t)(numSeeds ? numSeeds : 1) * 4);
  y.scalars = take(256);
  size_t s1 = 0, s2 = 0;
  (void)rocprim::exclusive_scan(nullptr, s1, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)numLists + 1, rocprim::plus<uint32_t>());
  (void)rocprim::exclusive_scan(nullptr, s2, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)(numSeeds ? numSeeds : 1), rocprim::plus<uint32_t>());
  y.scanTmpBytes = s1 > s2 ? s1 : s2;
  y.scanTmp = take(y.scanTmpBytes ? y.scanTmpBytes : 256);
  y.total = p;
  return y;
}
bool sizes_of(uint32_t numImages, const uint32_t* numFeatures, uint32_t* numLists, uint32_t* maxSeeds, uint32_t* numSeeds) {
  if (numImages < 2 || numImages > (uint32_t)kMaxImages || !numFeatures) return false;
  uint64_t lists = 0, seeds = 0;
  uint32_t mx = 0;
  for (uint32_t i = 0; i + 1 < numImages; ++i) lists += numFeatures[i];
  for (uint32_t i = 0; i + 2 < numImages; ++i) {
    seeds += numFeatures[i];
    mx = numFeatures[i] > mx ? numFeatures[i] : mx;
  }
  if (lists > 0x7fffffffull) return false;
  *numLists = (uint32_t)lists;
  *maxSeeds = mx;
  *numSeeds = (uint32_t)seeds;
  return true;
}

}  // namespace

extern "C" {

size_t ssrlcv_hip_merge_workspace_bytes(uint32_t numImages, const uint32_t* numFeatures_host, uint32_t totalPairs) {
  uint32_t numLists = 0, maxSeeds = 0, numSeeds = 0;
  if (!sizes_of(numImages, numFeatures_host, &numLists, &maxSeeds, &numSeeds)) return 0;
  return make_layout(numLists, totalPairs, maxSeeds, numSeeds).total;
}

int ssrlcv_hip_merge_matches(uint32_t numImages, const uint32_t* numFeatures_host, uint32_t numPairs, const uint32_t* pairCounts_host,
                             const ssrlcv_uint2_pair* pairs, void* workspace, size_t workspaceBytes, ssrlcv_multimatch* matches,
                             ssrlcv_uint2* members, uint32_t* counts, uint32_t* rounds_host, ssrlcv_stream_t stream) {
  uint32_t numLists = 0, maxSeeds = 0, numSeeds = 0;
  if (!sizes_of(numImages, numFeatures_host, &numLists, &maxSeeds, &numSeeds) || (numPairs && !pairCounts_host) || !workspace || !matches ||
      !members || !counts)
    return SSRLCV_ERR_INVALID_ARG;
  uint64_t total64 = 0;
  for (uint32_t k = 0; k < numPairs; ++k) total64 += pairCounts_host[k];
  if (total64 > 0x7fffffffull) return SSRLCV_ERR_CAPACITY;  // MultiMatch::index is an int (members <= 2 x pairs)
  const uint32_t total = (uint32_t)total64;
  if (total && !pairs) return SSRLCV_ERR_INVALID_ARG;
  const Layout y = make_layout(numLists, total, maxSeeds, numSeeds);
  if (workspaceBytes < y.total) return SSRLCV_ERR_WORKSPACE;
  const hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  uint32_t* mask = (uint32_t*)(ws + y.mask);
  uint32_t* start = (uint32_t*)(ws + y.start);
  uint32_t* minReader = (uint32_t*)(ws + y.minReader);
  uint32_t* minWriter = (uint32_t*)(ws + y.minWriter);
  uint8_t* state = (uint8_t*)(ws + y.state);
  uint8_t* outcome = (uint8_t*)(ws + y.outcome);
  uint32_t* good = (uint32_t*)(ws + y.good);
  uint32_t* memberCnt = (uint32_t*)(ws + y.members);
  uint32_t* mmOff = (uint32_t*)(ws + y.mmOff);
  uint32_t* memOff = (uint32_t*)(ws + y.memOff);
  int* bad = (int*)(ws + y.scalars);
  uint32_t* unresolved = (uint32_t*)(ws + y.scalars + 64);
  Lists L;
  L.V = numImages;
  L.base[0] = 0;
  for (uint32_t i = 0; i < numImages; ++i) {
    L.nf[i] = numFeatures_host[i];
    L.base[i + 1] = L.base[i] + (i + 1 < numImages ? numFeatures_host[i] : 0u);
  }
  L.start = start;
  L.len = (uint32_t*)(ws + y.len);
  L.entries = (const U2*)(ws + y.entries);
  if (rounds_host) *rounds_host = 0;

  // ---- lists
  SSRLCV_HIP_TRY(hipMemsetAsync(mask, 0, (size_t)numLists * 4, st));
  SSRLCV_HIP_TRY(hipMemsetAsync(bad, 0, 128, st));
  const unsigned eb = (total + 255) / 256, lb = (numLists + 255) / 256;
  if (total) hipLaunchKernelGGL(k_merge_masks, dim3(eb), dim3(256), 0, st, L, pairs, total, mask, bad);
  if (numLists) hipLaunchKernelGGL(k_merge_len, dim3(lb), dim3(256), 0, st, mask, numLists, L.len);
  {
    // start = exclusive scan of the lengths over numLists + 1 elements (the element behind the last list is never read as a length)
    size_t tmp = y.scanTmpBytes;
    SSRLCV_HIP_TRY(rocprim::exclusive_scan(ws + y.scanTmp, tmp, (const uint32_t*)L.len, start, 0u, (size_t)numLists + 1, rocprim::plus<uint32_t>(), st));
  }
  if (total) hipLaunchKernelGGL(k_merge_fill, dim3(eb), dim3(256), 0, st, L, pairs, total, (const uint32_t*)mask, (U2*)(ws + y.entries));
  int badHost = 0;
  SSRLCV_HIP_TRY(hipMemcpyAsync(&badHost, bad, sizeof(int), hipMemcpyDeviceToHost, st));
  SSRLCV_HIP_TRY(hipStreamSynchronize(st));
  if (badHost) return SSRLCV_ERR_INVALID_ARG;

  // ---- the walk, image by image (only images 0..V-3 seed multi-matches, :969)
  uint32_t seedBase = 0, roundsTotal = 0;
  for (uint32_t i = 0; i + 2 < numImages; ++i) {
    const uint32_t nf = numFeatures_host[i];
    const unsigned sb = (nf + 255) / 256;
    if (nf) {
      hipLaunchKernelGGL(k_merge_begin, dim3(sb), dim3(256), 0, st, L, i, state, outcome);
      const uint32_t later = L.base[i + 1];  // marks of lists of later images only
      for (int r = 0;; ++r) {
        SSRLCV_HIP_TRY(hipMemsetAsync(minReader + later, 0xff, (size_t)(numLists - later) * 4, st));
        SSRLCV_HIP_TRY(hipMemsetAsync(minWriter + later, 0xff, (size_t)(numLists - later) * 4, st));
        SSRLCV_HIP_TRY(hipMemsetAsync(unresolved, 0, 4, st));
        hipLaunchKernelGGL(k_merge_mark, dim3(sb), dim3(256), 0, st, L, i, (const uint8_t*)state, minReader, minWriter, unresolved);
        uint32_t left = 0;
        SSRLCV_HIP_TRY(hipMemcpyAsync(&left, unresolved, 4, hipMemcpyDeviceToHost, st));
        SSRLCV_HIP_TRY(hipStreamSynchronize(st));
        if (!left) break;
        if (r >= kMaxRounds) {
          hipLaunchKernelGGL(k_merge_sequential, dim3(1), dim3(1), 0, st, L, i, state, outcome);
          break;
        }
        hipLaunchKernelGGL(k_merge_ready, dim3(sb), dim3(256), 0, st, L, i, state, (const uint32_t*)minReader, (const uint32_t*)minWriter);
        hipLaunchKernelGGL(k_merge_commit, dim3(sb), dim3(256), 0, st, L, i, state, outcome);
        ++roundsTotal;
      }
      hipLaunchKernelGGL(k_merge_tally, dim3(sb), dim3(256), 0, st, L, i, (const uint8_t*)outcome, seedBase, good, memberCnt);
    }
    seedBase += nf;
  }
  // ---- output in (image, feature) order
  if (numSeeds) {
    size_t tmp = y.scanTmpBytes;
    SSRLCV_HIP_TRY(rocprim::exclusive_scan(ws + y.scanTmp, tmp, (const uint32_t*)good, mmOff, 0u, (size_t)numSeeds, rocprim::plus<uint32_t>(), st));
    tmp = y.scanTmpBytes;
    SSRLCV_HIP_TRY(rocprim::exclusive_scan(ws + y.scanTmp, tmp, (const uint32_t*)memberCnt, memOff, 0u, (size_t)numSeeds, rocprim::plus<uint32_t>(), st));
    seedBase = 0;
    for (uint32_t i = 0; i + 2 < numImages; ++i) {
      const uint32_t nf = numFeatures_host[i];
      if (nf)
        hipLaunchKernelGGL(k_merge_emit, dim3((nf + 255) / 256), dim3(256), 0, st, L, i, seedBase, (const uint32_t*)good, (const uint32_t*)mmOff,
                           (const uint32_t*)memOff, matches, members);
      seedBase += nf;
    }
  }
  hipLaunchKernelGGL(k_merge_totals, dim3(1), dim3(1), 0, st, (const uint32_t*)good, (const uint32_t*)memberCnt, (const uint32_t*)mmOff,
                     (const uint32_t*)memOff, numSeeds, counts);
  if (rounds_host) *rounds_host = roundsTotal;
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

}  // extern "C"
