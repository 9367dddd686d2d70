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
//   output  : one pass over the seeds with two running sums, (accepted ? 1 : 0) and (accepted ? members : 0), by decoupled
//             look-back (scan_lookback.h); the records are written from them directly.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <mutex>
#include "device_math.h"
#include "scan_lookback.h"
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
  const uint32_t V = L.V;
  if (p.a.x >= V - 1 || p.b.x >= V || p.b.x <= p.a.x || p.a.y >= L.nf[p.a.x] || p.b.y >= L.nf[p.b.x]) return;  // flagged by k_merge_masks
  const uint32_t l = L.list_of(p.a.x, p.a.y);
  const uint32_t slot = (uint32_t)__popc(mask[l] & ((1u << p.b.x) - 1u));
  entries[L.start[l] + slot] = U2{p.b.x, p.b.y};
}

// ---- the walk of every seed image in ONE persistent launch (round 4).  Round 3 launched mark / ready / commit per round and
// read the number of unresolved seeds back to the host after every mark (13-16 stream synchronisations per call); now
// the rounds are phases of one kernel separated by a grid-wide barrier, and the call is asynchronous.
//
// The grid is at most ONE block per CU (256 threads, no LDS to speak of: far inside what a CU admits), so every block
// becomes resident -- at once on an idle chip, as soon as other kernels' blocks retire otherwise (a waiting block sleeps,
// it holds nothing another kernel needs) -- and every block passes every barrier: the loop bounds and the `left` test are
// grid-uniform, so the grid always drains.  A PLAIN launch since round 5: hipLaunchCooperativeKernel buys only the
// launch-time check of the grid against the occupancy query (MI355X_MICROARCH.md, coop-launch), and on this driver stack
// a process that has made one cooperative launch slows every OTHER process on the same GPU that has made one too -- two
// such processes are time-sliced against each other: the C++ flow ran 45 ms alone and 98 ms as a child of a Python process
// that had merged once (tools/flow_diag.py, profiles/r05_coop_launch.txt).
// Barrier = __threadfence (release: this thread's stores and atomics are visible device-wide, across the XCDs' L2s),
// one atomicAdd per block on a monotone counter, spin until it reaches (barriers so far) x gridDim.x, __threadfence
// (acquire: drop what this CU cached before the barrier).
struct WalkCtl {
  uint32_t* barrier;      // [0] groups arrived (monotone); [64 (1 + g)] arrivals of group g, each on its own 256-byte line
  uint32_t* unresolved;   // [(V - 2) x (kMaxRounds + 2)] one slot per (image, round), zeroed before the launch
  uint32_t* counts;       // counts[3] receives the number of rounds
  int* bad;               // set by k_merge_masks on malformed input: the walk is skipped; bit 2 (value 4) = a grid barrier timed out (below)
};
// Same-address atomics serialise at ~50 ns each on this part, so 256 blocks arriving on ONE counter cost 13 us per
// barrier (the first version: 60 barriers = 3.7 ms for a merge whose kernels take 0.4).  Arrival is two-level: a block
// arrives on the counter of its group (blockIdx mod kBarrierGroups, 16 arrivals each, the groups in parallel), the last
// of a group on the global one, everybody polls the global one.  Fences: thread 0 alone releases before and acquires
// after -- the block's other threads are ordered with it by the __syncthreads on either side, and an agent-scope
// release / acquire writes back / invalidates the caches for the whole CU, not for one wave.
constexpr uint32_t kBarrierGroups = 16;
constexpr size_t kBarrierBytes = 256 * (1 + kBarrierGroups);
// The launch is a plain one (a cooperative launch makes processes sharing the GPU time-slice against each other, round 5), so
// nothing but the host-side sizing (one block per CU, at most what the occupancy query grants) guarantees that every block
// is resident.  Should a block ever not be (CU masking, another persistent kernel holding the CUs), the spin is BOUNDED:
// after ~2^22 polls (a few hundred milliseconds; a barrier normally takes microseconds) the waiter sets bit 2 of *bad and every
// block leaves -- the call returns SSRLCV_ERR_UNSUPPORTED instead of hanging the GPU.  -> false: the walk is abandoned.
constexpr uint32_t kBarrierSpinLimit = 1u << 22;
__device__ __forceinline__ bool grid_barrier(uint32_t* ctr, uint32_t& passed, int* bad) {
  __shared__ int s_abort;
  __syncthreads();
  ++passed;
  if (threadIdx.x == 0) {
    __threadfence();
    const uint32_t groups = gridDim.x < kBarrierGroups ? gridDim.x : kBarrierGroups;
    const uint32_t g = blockIdx.x % kBarrierGroups;
    const uint32_t groupSize = (gridDim.x - g + kBarrierGroups - 1) / kBarrierGroups;  // blocks with this residue
    if (atomicAdd(ctr + 64 * (1 + g), 1u) == passed * groupSize - 1u) atomicAdd(ctr, 1u);
    const uint32_t target = passed * groups;
    int abort = 0;
    for (uint32_t spins = 0; __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++spins) {
      __builtin_amdgcn_s_sleep(1);
      if ((spins & 1023u) == 1023u && (__hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4)) { abort = 1; break; }
      if (spins >= kBarrierSpinLimit) {
        atomicOr(bad, 4);
        abort = 1;
        break;
      }
    }
    s_abort = abort;
    __threadfence();
  }
  __syncthreads();
  return s_abort == 0;
}
// state[f]: 1 = unresolved, 2 = ready this round
__global__ __launch_bounds__(256) void k_merge_walk(Lists L, uint32_t numLists, uint8_t* __restrict__ state, uint8_t* __restrict__ outcome,
                                                    uint32_t* __restrict__ minReader, uint32_t* __restrict__ minWriter,
                                                    uint32_t* __restrict__ good, uint32_t* __restrict__ members, WalkCtl ctl) {
  if (*ctl.bad) return;  // grid-uniform (written by an earlier kernel)
  const uint32_t tid = blockIdx.x * 256 + threadIdx.x, nthreads = gridDim.x * 256;
  uint32_t passed = 0, seedBase = 0, rounds = 0;
  for (uint32_t i = 0; i + 2 < L.V; ++i) {
    const uint32_t nf = L.nf[i];
    const uint32_t later = L.base[i + 1];  // marks of lists of later images only
    for (uint32_t f = tid; f < nf; f += nthreads) {
      state[f] = L.len[L.list_of(i, f)] != 0;
      outcome[f] = kSkip;
    }
    for (uint32_t l = later + tid; l < numLists; l += nthreads) minReader[l] = minWriter[l] = 0xffffffffu;
    if (!grid_barrier(ctl.barrier, passed, ctl.bad)) return;
    for (uint32_t r = 0;; ++r) {
      // mark: R(f) and W(f) of every unresolved seed on the current state
      uint32_t* slot = ctl.unresolved + i * (kMaxRounds + 2) + (r < (uint32_t)kMaxRounds + 1 ? r : (uint32_t)kMaxRounds + 1);
      uint32_t mine = 0;
      for (uint32_t f = tid; f < nf; f += nthreads) {
        if (!state[f]) continue;
        const uint32_t a = L.list_of(i, f);
        walk(L, a, [&](uint32_t l) { atomicMin(&minReader[l], f); });
        for_each_cleared(L, a, [&](uint32_t l) { atomicMin(&minWriter[l], f); });
        ++mine;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
      if ((threadIdx.x & 63) == 0 && mine) atomicAdd(slot, mine);
      if (!grid_barrier(ctl.barrier, passed, ctl.bad)) return;
      const uint32_t left = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the same for every thread
      if (!left) break;
      if (r >= (uint32_t)kMaxRounds) {  // what is left after kMaxRounds, in upstream's order, by one thread
        if (tid == 0) {
          for (uint32_t f = 0; f < nf; ++f) {
            if (!state[f]) continue;
            const uint32_t a = L.list_of(i, f);
            const Outcome o = walk(L, a, [](uint32_t) {});
            outcome[f] = o;
            commit(L, a, o);
            state[f] = 0;
          }
        }
        if (!grid_barrier(ctl.barrier, passed, ctl.bad)) return;
        break;
      }
      // ready: no unresolved lower seed clears what f reads, none reads what f clears (the marks of f itself are f)
      for (uint32_t f = tid; f < nf; f += nthreads) {
        if (!state[f]) continue;
        const uint32_t a = L.list_of(i, f);
        bool ready = true;
        walk(L, a, [&](uint32_t l) { ready = ready && minWriter[l] >= f; });
        for_each_cleared(L, a, [&](uint32_t l) { ready = ready && minReader[l] >= f; });
        if (ready) state[f] = 2;
      }
      if (!grid_barrier(ctl.barrier, passed, ctl.bad)) return;
      // commit the ready seeds (independent of each other and of every earlier unresolved seed), reset the marks
      for (uint32_t f = tid; f < nf; f += nthreads) {
        if (state[f] != 2) continue;
        const uint32_t a = L.list_of(i, f);
        const Outcome o = walk(L, a, [](uint32_t) {});
        outcome[f] = o;
        commit(L, a, o);
        state[f] = 0;
      }
      for (uint32_t l = later + tid; l < numLists; l += nthreads) minReader[l] = minWriter[l] = 0xffffffffu;
      ++rounds;
      if (!grid_barrier(ctl.barrier, passed, ctl.bad)) return;
    }
    // tally: an accepted seed's own list is final (only seeds of earlier images clear it)
    for (uint32_t f = tid; f < nf; f += nthreads) {
      const bool g = outcome[f] == kGood;
      good[seedBase + f] = g ? 1u : 0u;
      members[seedBase + f] = g ? L.len[L.list_of(i, f)] + 1u : 0u;
    }
    seedBase += nf;
    if (i + 3 < L.V && !grid_barrier(ctl.barrier, passed, ctl.bad)) return;  // the next image reuses state / outcome
  }
  if (tid == 0) ctl.counts[3] = rounds;
}
// ---- output in (image, feature) order: one pass over the seeds (seed index = list index: image i's seeds are its
// lists) with two running sums by decoupled look-back (scan_lookback.h) -- multi-matches before this seed, members before
// it -- and the records written straight from them.  (Round 3: two library scans, one emit launch per image, a totals
// kernel.)
constexpr int kEmitItems = 4;
__global__ __launch_bounds__(svs::kThreads) void k_merge_emit(Lists L, uint32_t numSeeds, const uint32_t* __restrict__ good,
                                                              const uint32_t* __restrict__ memberCnt, ssrlcv_multimatch* __restrict__ mm,
                                                              ssrlcv_uint2* __restrict__ mem, uint32_t* __restrict__ counts,
                                                              const int* __restrict__ bad, svs::TileScan<2> ts) {
  if (*bad) {  // grid-uniform: malformed input, the walk did not run
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      counts[0] = counts[1] = counts[3] = 0u;
      counts[2] = (uint32_t)*bad;
    }
    return;
  }
  constexpr uint32_t kTile = svs::kThreads * kEmitItems;
  for (uint32_t tile = svs::next_tile(ts.counter); tile < ts.numTiles; tile = svs::next_tile(ts.counter)) {
    const uint32_t base = tile * kTile + threadIdx.x * kEmitItems;
    uint32_t g[kEmitItems], c[kEmitItems], mine[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < kEmitItems; ++j) {
      const bool in = base + j < numSeeds;
      g[j] = in ? good[base + j] : 0u;
      c[j] = in ? memberCnt[base + j] : 0u;
      mine[0] += g[j];
      mine[1] += c[j];
    }
    uint32_t excl[2], total[2], prefix[2];
    svs::block_exclusive<2>(mine, excl, total);
    svs::tile_prefix<2>(ts, tile, total, prefix);
    uint32_t mmAt = prefix[0] + excl[0], memAt = prefix[1] + excl[1];
#pragma unroll
    for (int j = 0; j < kEmitItems; ++j) {
      if (g[j]) {
        const uint32_t a = base + j, n = L.len[a];
        uint32_t i = 0;
        while (i + 1 < L.V && L.base[i + 1] <= a) ++i;
        ssrlcv_multimatch m;
        m.numKeyPoints = n + 1;
        m.index = (int)memAt;
        mm[mmAt] = m;
        mem[memAt].x = i;
        mem[memAt].y = a - L.base[i];
        for (uint32_t k = 0; k < n; ++k) {
          const U2 e = L.entries[L.start[a] + k];
          mem[memAt + 1 + k].x = e.x;
          mem[memAt + 1 + k].y = e.y;
        }
        ++mmAt;
        memAt += c[j];
      }
    }
    if (tile == ts.numTiles - 1 && threadIdx.x == 0) {
      counts[0] = prefix[0] + total[0];
      counts[1] = prefix[1] + total[1];
      counts[2] = 0u;  // counts[3] (rounds) was written by k_merge_walk
    }
  }
}
__global__ void k_merge_no_seeds(const int* __restrict__ bad, uint32_t* __restrict__ counts) {  // two images: nothing seeds a multi-match
  counts[0] = counts[1] = counts[3] = 0u;
  counts[2] = (uint32_t)*bad;
}
inline uint32_t emit_tiles(uint32_t numSeeds) { return (numSeeds + svs::kThreads * kEmitItems - 1) / (svs::kThreads * kEmitItems); }

inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }
struct Layout {
  size_t mask, len, start, entries, minReader, minWriter, state, outcome, good, members, scalars, ctl, scanTmp, total;
  size_t ctlBytes;
  size_t scanTmpBytes;
};
Layout make_layout(uint32_t numImages, uint32_t numLists, uint32_t total, uint32_t maxSeeds, uint32_t numSeeds) {
  Layout y;
  size_t p = 0;
  auto take = [&](size_t bytes) { const size_t at = p; p += up256(bytes); return at; };
  y.mask = take((size_t)numLists * 4);
  y.len = take(((size_t)numLists + 1) * 4);  // + the element the scan over numLists + 1 inputs reads (zeroed)
  y.start = take(((size_t)numLists + 1) * 4);
  y.entries = take((size_t)(total ? total : 1) * 8);
  y.minReader = take((size_t)numLists * 4);
  y.minWriter = take((size_t)numLists * 4);
  y.state = take(maxSeeds ? maxSeeds : 1);
  y.outcome = take(maxSeeds ? maxSeeds : 1);
  y.good = take((size_t)(numSeeds ? numSeeds : 1) * 4);
  y.members = take((size_t)(numSeeds ? numSeeds : 1) * 4);
  y.scalars = take(256);
  // barrier counters + one unresolved-count slot per (seed image, round)
  y.ctlBytes = kBarrierBytes + (size_t)(numImages > 2 ? numImages - 2 : 1) * (kMaxRounds + 2) * 4;
  y.ctl = take(y.ctlBytes);
  // descriptors of the two single-pass scans (used one after the other): list starts over numLists + 1 lengths, emit over the seeds
  const size_t s1 = svs::workspace_bytes<1>(svs::scan_tiles<8>(numLists + 1));
  const size_t s2 = svs::workspace_bytes<2>(emit_tiles(numSeeds ? numSeeds : 1));
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
  return make_layout(numImages, numLists, totalPairs, maxSeeds, numSeeds).total;
}

int ssrlcv_hip_merge_matches(uint32_t numImages, const uint32_t* numFeatures_host, uint32_t numPairs, const uint32_t* pairCounts_host,
                             const ssrlcv_uint2_pair* pairs, void* workspace, size_t workspaceBytes, ssrlcv_multimatch* matches,
                             ssrlcv_uint2* members, uint32_t* counts, ssrlcv_stream_t stream) {
  uint32_t numLists = 0, maxSeeds = 0, numSeeds = 0;
  if (!sizes_of(numImages, numFeatures_host, &numLists, &maxSeeds, &numSeeds) || (numPairs && !pairCounts_host) || !workspace || !matches ||
      !members || !counts)
    return SSRLCV_ERR_INVALID_ARG;
  uint64_t total64 = 0;
  for (uint32_t k = 0; k < numPairs; ++k) total64 += pairCounts_host[k];
  if (total64 > 0x7fffffffull) return SSRLCV_ERR_CAPACITY;  // MultiMatch::index is an int (members <= 2 x pairs)
  const uint32_t total = (uint32_t)total64;
  if (total && !pairs) return SSRLCV_ERR_INVALID_ARG;
  const Layout y = make_layout(numImages, numLists, total, maxSeeds, numSeeds);
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
  int* bad = (int*)(ws + y.scalars);
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

  // ---- lists
  SSRLCV_HIP_TRY(hipMemsetAsync(mask, 0, (size_t)numLists * 4, st));
  SSRLCV_HIP_TRY(hipMemsetAsync(bad, 0, 128, st));
  SSRLCV_HIP_TRY(hipMemsetAsync(ws + y.ctl, 0, y.ctlBytes, st));
  SSRLCV_HIP_TRY(hipMemsetAsync(L.len + numLists, 0, 4, st));  // the scan's last input (its result is the total)
  const unsigned eb = (total + 255) / 256, lb = (numLists + 255) / 256;
  if (total) hipLaunchKernelGGL(k_merge_masks, dim3(eb), dim3(256), 0, st, L, pairs, total, mask, bad);
  if (numLists) hipLaunchKernelGGL(k_merge_len, dim3(lb), dim3(256), 0, st, mask, numLists, L.len);
  // start = exclusive scan of the lengths over numLists + 1 elements (the last input is zero, its result the total)
  SSRLCV_HIP_TRY(svs::exclusive_scan<8>((const uint32_t*)L.len, start, numLists + 1, ws + y.scanTmp, st));
  if (total) hipLaunchKernelGGL(k_merge_fill, dim3(eb), dim3(256), 0, st, L, pairs, total, (const uint32_t*)mask, (U2*)(ws + y.entries));

  // ---- the walk, image by image (only images 0..V-3 seed multi-matches, :969), in one persistent launch (at most one block per CU)
  if (numSeeds) {
    // co-resident blocks of k_merge_walk, per device (a process may drive devices of different size or partition mode)
    static std::mutex s_mu;
    static std::map<int, int> s_blocksOf;
    int dev = 0, s_blocks = 0;
    SSRLCV_HIP_TRY(hipGetDevice(&dev));
    {
      std::lock_guard<std::mutex> lock(s_mu);
      auto it = s_blocksOf.find(dev);
      if (it == s_blocksOf.end()) {
        int cus = 0, perCu = 0;
        SSRLCV_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        SSRLCV_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_merge_walk, 256, 0));
        if (cus <= 0 || perCu <= 0) return SSRLCV_ERR_UNSUPPORTED;
        // one block per CU (more only lengthen the barrier), never more than the occupancy query grants in all
        it = s_blocksOf.emplace(dev, cus < cus * perCu ? cus : cus * perCu).first;
      }
      s_blocks = it->second;
    }
    unsigned blocks = (maxSeeds + 255) / 256;
    if (blocks > (unsigned)s_blocks) blocks = (unsigned)s_blocks;
    if (blocks == 0) blocks = 1;
    WalkCtl ctl;
    ctl.barrier = (uint32_t*)(ws + y.ctl);
    ctl.unresolved = (uint32_t*)(ws + y.ctl + kBarrierBytes);
    ctl.counts = counts;
    ctl.bad = bad;
    hipLaunchKernelGGL(k_merge_walk, dim3(blocks), dim3(256), 0, st, L, numLists, state, outcome, minReader, minWriter, good, memberCnt, ctl);
  }
  // ---- output in (image, feature) order
  if (numSeeds) {
    const uint32_t tiles = emit_tiles(numSeeds);
    SSRLCV_HIP_TRY(hipMemsetAsync(ws + y.scanTmp, 0, svs::workspace_bytes<2>(tiles), st));
    const svs::TileScan<2> ts = svs::make_tile_scan<2>(ws + y.scanTmp, tiles);
    hipLaunchKernelGGL(k_merge_emit, dim3(tiles < 2048u ? tiles : 2048u), dim3(svs::kThreads), 0, st, L, numSeeds, (const uint32_t*)good,
                       (const uint32_t*)memberCnt, matches, members, counts, (const int*)bad, ts);
  } else {
    hipLaunchKernelGGL(k_merge_no_seeds, dim3(1), dim3(1), 0, st, (const int*)bad, counts);
  }
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

}  // extern "C"
