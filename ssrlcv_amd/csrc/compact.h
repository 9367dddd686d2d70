// ssrlcv_amd/csrc/compact.h -- order-preserving multi-block stream compaction / partition for gfx950.
//
// Replaces the thrust::remove / remove_if / copy_if / stable_sort-by-small-key call sites of the hot path
// (src/FeatureFactory.cu:126,189,250,595,597; src/MatchFactory.cu:37,63,89).  Three launches:
//   1. k_count  : each WAVE owns a run of 64 * PER_THREAD consecutive elements and counts, per key, how many of them
//                 carry that key (wave64 ballots).  An element may carry several keys (bit mask), e.g. a pixel that is
//                 an extremum of two DoG levels; mask 0 removes the element;
//   2. k_scan   : one block turns the [key][wave] count table into exclusive offsets in key-major order, so the
//                 output is grouped by key and, inside a key, keeps the input order (= stable partition);
//   3. k_scatter: each wave walks its run again, ranks its elements with ballots + popcounts and emits them.
// Everything is wave-granular: no LDS, no barriers (the first version exchanged per-round counts through LDS with three
// __syncthreads per 256 elements, which was 0.4 ms of the 67 Mpx extrema compaction).
// The mask functor must be pure (it is evaluated in passes 1 and 3).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svc {

constexpr int kThreads = 256;
constexpr int kWavesPerBlock = kThreads / 64;

__host__ __device__ inline uint32_t num_chunks(uint32_t n, int perThread) {
  uint32_t chunk = (uint32_t)kThreads * (uint32_t)perThread;
  return (n + chunk - 1) / chunk;
}

// counts layout: counts[key * numWaves + wave], wave = block * 4 + wave-in-block; numRuns = numBlocks * 4
// `live` (nullable): device word holding the number of leading elements that can carry a key at all (elements past
// *live x liveScale all have mask 0, the caller guarantees it).  The key-point lists are allocated for a capacity that is
// 5..20 times their usual length; with it the passes touch only the runs below that bound -- the single-block scan was
// walking 40 960 table entries for a list that fills 2 000 of them.
__device__ __forceinline__ uint32_t live_elements(const int* live, uint32_t liveScale, uint32_t n) {
  if (!live) return n;
  const int v = *live;
  const unsigned long long e = (unsigned long long)(v > 0 ? v : 0) * liveScale;
  return e < n ? (uint32_t)e : n;
}

// Bookkeeping that has to follow a partition (it rewrites the list's length / segment table, which the scatter's mask
// functor still reads) used to be a one-thread kernel of its own after every partition: ~6 us each, five per octave
// chain, all on the critical path of the short octaves.  It now runs in the LAST block of the scatter kernel to finish
// (a block counter next to the totals, zeroed by the scan): every other block has then evaluated its masks.
struct NoPost {
  __device__ void operator()(const uint32_t*) const {}
};
// Only the `participants` leading blocks count themselves (same-address atomics serialise at ~50 ns each: all 2 048
// blocks of a capacity-sized launch would cost more than the kernel this replaces); the blocks behind them are past
// the live bound and touch nothing the bookkeeping changes.
template <typename T> struct is_no_post { static constexpr bool value = false; };
template <typename PostFn>
__device__ __forceinline__ void last_block_post(uint32_t* done, const uint32_t* totals, uint32_t participants, PostFn post) {
  if (is_no_post<PostFn>::value) return;
  if (participants == 0) participants = 1;
  if (blockIdx.x >= participants) return;  // block-uniform
  __shared__ bool s_last;
  __syncthreads();  // the block's emits are issued
  if (threadIdx.x == 0) {
    __threadfence();
    s_last = atomicAdd(done, 1u) == participants - 1;
  }
  __syncthreads();
  if (s_last && threadIdx.x == 0) {
    __threadfence();
    post(totals);
  }
}

template <> struct is_no_post<NoPost> { static constexpr bool value = true; };
template <typename PostFn>
__global__ void k_post_only(const uint32_t* totals, PostFn post) { post(totals); }

template <int NKEYS, int PER_THREAD, typename MaskFn>
__global__ __launch_bounds__(kThreads) void k_count(uint32_t n, uint32_t numRuns, MaskFn maskfn,
                                                    uint32_t* __restrict__ counts, const int* live, uint32_t liveScale) {
  const unsigned lane = threadIdx.x & 63;
  const uint32_t run = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const uint32_t base = run * (64u * PER_THREAD);
  if (base >= live_elements(live, liveScale, n)) {  // wave-uniform
    // with a live bound the scan does not read this run's counts; without one it reads every run of the table
    if (!live && lane < NKEYS) counts[lane * numRuns + run] = 0;
    return;
  }
  uint32_t cnt[NKEYS];
#pragma unroll
  for (int k = 0; k < NKEYS; ++k) cnt[k] = 0;
  for (int r = 0; r < PER_THREAD; ++r) {
    if (base + r * 64u >= n) break;  // wave-uniform
    uint32_t i = base + r * 64u + lane;
    uint32_t mask = (i < n) ? maskfn(i) : 0u;
#pragma unroll
    for (int k = 0; k < NKEYS; ++k) cnt[k] += (uint32_t)__popcll(__ballot((mask >> k) & 1u));
  }
  if (lane < NKEYS) {
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < NKEYS; ++k) v = lane == (unsigned)k ? cnt[k] : v;
    counts[lane * numRuns + run] = v;
  }
}

// Exclusive scan over the NKEYS*numBlocks table (numBlocks = number of wave runs here) in key-major order; totals[k] = number of elements with key k,
// totals[NKEYS] = grand total.  Single block, each thread scanning 4 consecutive entries per round: 1024 threads for the
// large pixel-domain tables, 256 for the list partitions -- a 16-wave block needs 16 free wave slots on ONE CU, and the
// short chains of octaves 1-3 waited up to 0.96 ms for that beside the oversubscribed orientation kernel of octave 0.
template <int NKEYS, int THREADS>
__global__ __launch_bounds__(THREADS) void k_scan(uint32_t numBlocks, uint32_t* __restrict__ counts,
                                                  uint32_t* __restrict__ totals, uint32_t liveRunsOr0, const int* live,
                                                  uint32_t liveScale, uint32_t runLen, uint32_t n) {
  __shared__ uint32_t s_wave[THREADS / 64];
  __shared__ uint32_t s_carry;
  __shared__ uint32_t s_keystart[NKEYS + 1];
  // runs that were counted: all of them, or those below the live bound
  uint32_t lr = numBlocks;
  if (live) {
    const uint32_t le = live_elements(live, liveScale, n);
    lr = (le + runLen - 1) / runLen;
    if (lr > numBlocks) lr = numBlocks;
  }
  (void)liveRunsOr0;
  if (threadIdx.x == 0) totals[NKEYS + 1] = 0;  // block counter of the scatter kernel (last_block_post)
  if (lr == 0) {
    if (threadIdx.x <= NKEYS) totals[threadIdx.x] = 0;
    return;
  }
  const uint32_t total_entries = NKEYS * lr;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (uint32_t start = 0; start < total_entries; start += THREADS * 4) {
    uint32_t i0 = start + threadIdx.x * 4;
    uint32_t v[4], addr[4], runOf[4], keyOf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t i = i0 + j;
      keyOf[j] = i / lr;
      runOf[j] = i - keyOf[j] * lr;
      addr[j] = keyOf[j] * numBlocks + runOf[j];
      v[j] = (i < total_entries) ? counts[addr[j]] : 0;
    }
    uint32_t tsum = v[0] + v[1] + v[2] + v[3];
    uint32_t x = tsum;  // inclusive scan of per-thread sums inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t y = __shfl_up(x, o, 64);
      if ((threadIdx.x & 63) >= (unsigned)o) x += y;
    }
    if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = x;
    __syncthreads();
    uint32_t wave_off = 0;
    for (unsigned w = 0; w < (threadIdx.x >> 6); ++w) wave_off += s_wave[w];
    uint32_t excl = s_carry + wave_off + x - tsum;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t i = i0 + j;
      if (i < total_entries) {
        counts[addr[j]] = excl;
        if (runOf[j] == 0) s_keystart[keyOf[j]] = excl;
      }
      excl += v[j];
    }
    __syncthreads();
    if (threadIdx.x == THREADS - 1) s_carry = excl;
    __syncthreads();
  }
  if (threadIdx.x == 0) s_keystart[NKEYS] = s_carry;
  __syncthreads();
  if (threadIdx.x < NKEYS) totals[threadIdx.x] = s_keystart[threadIdx.x + 1] - s_keystart[threadIdx.x];
  if (threadIdx.x == 0) totals[NKEYS] = s_carry;
}

// emit(i, key, dst): writes input element i (as a member of `key`) to output slot dst.
template <int NKEYS, int PER_THREAD, typename MaskFn, typename EmitFn, typename PostFn>
__global__ __launch_bounds__(kThreads) void k_scatter(uint32_t n, uint32_t numRuns, MaskFn maskfn, EmitFn emit,
                                                      const uint32_t* __restrict__ offsets, const int* live, uint32_t liveScale,
                                                      uint32_t* totals, PostFn post) {
  const unsigned lane = threadIdx.x & 63;
  const uint32_t run = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const uint32_t base = run * (64u * PER_THREAD);
  if (base < live_elements(live, liveScale, n)) {  // wave-uniform
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t off[NKEYS];  // running output offset per key of this wave's run (wave-uniform)
#pragma unroll
    for (int k = 0; k < NKEYS; ++k) off[k] = offsets[k * numRuns + run];
    for (int r = 0; r < PER_THREAD; ++r) {
      if (base + r * 64u >= n) break;  // wave-uniform
      uint32_t i = base + r * 64u + lane;
      uint32_t mask = (i < n) ? maskfn(i) : 0u;
      if (__ballot(mask != 0u) == 0ull) continue;  // nothing kept in these 64 elements (the common case for pixels)
#pragma unroll
      for (int k = 0; k < NKEYS; ++k) {
        const bool has = (mask >> k) & 1u;
        const unsigned long long m = __ballot(has);
        if (has) emit(i, k, off[k] + (uint32_t)__popcll(m & below));
        off[k] += (uint32_t)__popcll(m);
      }
    }
  }
  const uint32_t le = live_elements(live, liveScale, n);
  last_block_post(totals + NKEYS + 1, totals, (le + kThreads * PER_THREAD - 1) / (kThreads * PER_THREAD), post);
}

// Workspace (in uint32 words): NKEYS * numRuns counts + (NKEYS + 1) totals + the scatter kernel's block counter.
template <int NKEYS, int PER_THREAD>
inline size_t workspace_words(uint32_t n) {
  return (size_t)NKEYS * kWavesPerBlock * num_chunks(n, PER_THREAD) + NKEYS + 2;
}

// Launches the three passes over elements [0, n).  *totals_out = device pointer to NKEYS + 1 words in the workspace.
// post(totals): device functor run once, after every element has been emitted (see last_block_post)
template <int NKEYS, int PER_THREAD, typename MaskFn, typename EmitFn, typename PostFn = NoPost>
inline hipError_t partition(uint32_t n, MaskFn maskfn, EmitFn emit, uint32_t* workspace, uint32_t** totals_out,
                            hipStream_t stream, const int* live = nullptr, uint32_t liveScale = 1, PostFn post = PostFn()) {
  uint32_t nb = num_chunks(n, PER_THREAD);
  uint32_t runs = nb * kWavesPerBlock;
  uint32_t* counts = workspace;
  uint32_t* totals = workspace + (size_t)NKEYS * runs;
  if (totals_out) *totals_out = totals;
  if (nb == 0) {  // nothing to partition: totals are zero, the bookkeeping still has to happen
    hipError_t e0 = hipMemsetAsync(totals, 0, sizeof(uint32_t) * (NKEYS + 2), stream);
    if (e0 != hipSuccess) return e0;
    hipLaunchKernelGGL((k_post_only<PostFn>), dim3(1), dim3(1), 0, stream, totals, post);
    return hipGetLastError();
  }
  const uint32_t runLen = 64u * PER_THREAD;
  hipLaunchKernelGGL((k_count<NKEYS, PER_THREAD, MaskFn>), dim3(nb), dim3(kThreads), 0, stream, n, runs, maskfn, counts, live,
                     liveScale);
  // a list partition with a live bound scans a short table: the 256-thread block is enough (and does not wait for 16
  // free wave slots on one CU)
  if (!live && (size_t)NKEYS * runs > 16384)
    hipLaunchKernelGGL((k_scan<NKEYS, 1024>), dim3(1), dim3(1024), 0, stream, runs, counts, totals, 0u, live, liveScale, runLen, n);
  else
    hipLaunchKernelGGL((k_scan<NKEYS, 256>), dim3(1), dim3(256), 0, stream, runs, counts, totals, 0u, live, liveScale, runLen, n);
  hipLaunchKernelGGL((k_scatter<NKEYS, PER_THREAD, MaskFn, EmitFn, PostFn>), dim3(nb), dim3(kThreads), 0, stream, n, runs,
                     maskfn, emit, counts, live, liveScale, totals, post);
  return hipGetLastError();
}

// ---- byte-flag variant ("lane = round") ------------------------------------------------------------------------------
// The pixel-domain partition reads one flag BYTE per element (bit k = element carries key k) and keeps well under one
// element in a hundred.  With one element per lane and round, a wave's run of 4096 elements cost 64 dependent byte loads
// and 64 x NKEYS ballots.  Here lane l loads the 64 bytes of "its" round [base + 64 l, base + 64 l + 64) as four
// 16-byte words and squeezes bit k of every byte into a 64-bit mask per key (16 x {shift, and, multiply, extract}):
// per-key counts are popcounts, output slots an exclusive wave prefix over the lanes plus the rank of the bit inside
// the lane's mask -- the same key-major, input-ordered output as the generic kernels, from 4 loads and ~250
// instructions per 4096 elements.  Requirements: n % 64 == 0, `flags` 16-byte aligned, shift + NKEYS <= 8.
constexpr int kFlagRun = 4096;  // elements per wave

// shift: key k is bit shift + k of the flag byte
template <int NKEYS>
__device__ __forceinline__ void flag_masks(const uint8_t* __restrict__ flags, uint32_t base, uint32_t n, unsigned lane, int shift,
                                           unsigned long long (&m)[NKEYS]) {
#pragma unroll
  for (int k = 0; k < NKEYS; ++k) m[k] = 0ull;
  const uint32_t first = base + 64u * lane;
  if (first >= n) return;
  const uint4* src = reinterpret_cast<const uint4*>(flags + first);
  uint32_t w[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint4 v = src[q];
    w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
  }
#pragma unroll
  for (int k = 0; k < NKEYS; ++k) {
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int d = 0; d < 16; ++d) {
      // bit k of the four bytes -> bits 21..24 of the product (1 + 2^7 + 2^14 + 2^21: every target bit gets one term)
      const uint32_t nib = ((((w[d] >> (k + shift)) & 0x01010101u) * 0x00204081u) >> 21) & 0xFu;
      if (d < 8) lo |= nib << (4 * d);
      else hi |= nib << (4 * (d - 8));
    }
    m[k] = ((unsigned long long)hi << 32) | lo;
  }
}

template <int NKEYS>
__global__ __launch_bounds__(kThreads) void k_count_flags(uint32_t n, uint32_t numRuns, const uint8_t* __restrict__ flags, int shift,
                                                          uint32_t* __restrict__ counts) {
  const unsigned lane = threadIdx.x & 63;
  const uint32_t run = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  unsigned long long m[NKEYS];
  flag_masks<NKEYS>(flags, run * (uint32_t)kFlagRun, n, lane, shift, m);
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < NKEYS; ++k) {
    uint32_t c = (uint32_t)__popcll(m[k]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    mine = lane == (unsigned)k ? c : mine;
  }
  if (lane < NKEYS) counts[lane * numRuns + run] = mine;
}

template <int NKEYS, typename EmitFn, typename PostFn>
__global__ __launch_bounds__(kThreads) void k_scatter_flags(uint32_t n, uint32_t numRuns, const uint8_t* __restrict__ flags, int shift,
                                                            EmitFn emit, const uint32_t* __restrict__ offsets, uint32_t* totals,
                                                            PostFn post) {
  const unsigned lane = threadIdx.x & 63;
  const uint32_t run = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const uint32_t base = run * (uint32_t)kFlagRun;
  unsigned long long m[NKEYS];
  flag_masks<NKEYS>(flags, base, n, lane, shift, m);
#pragma unroll
  for (int k = 0; k < NKEYS; ++k) {
    const uint32_t c = (uint32_t)__popcll(m[k]);
    uint32_t incl = c;  // inclusive prefix over the lanes (= rounds, in element order)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(incl, o, 64);
      if (lane >= (unsigned)o) incl += y;
    }
    uint32_t dst = offsets[k * numRuns + run] + incl - c;
    unsigned long long bits = m[k];
    while (__ballot(bits != 0ull) != 0ull) {  // as many rounds as the fullest lane has elements of this key
      if (bits != 0ull) {
        const int j = __ffsll((long long)bits) - 1;
        bits &= bits - 1ull;
        emit(base + 64u * lane + (uint32_t)j, k, dst++);
      }
    }
  }
  last_block_post(totals + NKEYS + 1, totals, gridDim.x, post);
}

// partition() for byte flags; falls back to the generic kernels when the fast path's requirements do not hold.
template <int NKEYS, typename EmitFn, typename PostFn = NoPost>
inline hipError_t partition_flags(uint32_t n, const uint8_t* flags, int shift, EmitFn emit, uint32_t* workspace, uint32_t** totals_out,
                                  hipStream_t stream, PostFn post = PostFn()) {
  if (n % 64u != 0u || (reinterpret_cast<size_t>(flags) & 15u) != 0u) {
    auto maskfn = [=] __device__(uint32_t i) -> uint32_t { return ((uint32_t)flags[i] >> shift) & ((1u << NKEYS) - 1u); };
    return partition<NKEYS, kFlagRun / 64>(n, maskfn, emit, workspace, totals_out, stream, nullptr, 1u, post);
  }
  const uint32_t runs0 = (n + kFlagRun - 1) / kFlagRun;
  const uint32_t nb = (runs0 + kWavesPerBlock - 1) / kWavesPerBlock;
  const uint32_t runs = nb * kWavesPerBlock;
  uint32_t* counts = workspace;
  uint32_t* totals = workspace + (size_t)NKEYS * runs;
  if (totals_out) *totals_out = totals;
  if (nb == 0) {
    hipError_t e0 = hipMemsetAsync(totals, 0, sizeof(uint32_t) * (NKEYS + 2), stream);
    if (e0 != hipSuccess) return e0;
    hipLaunchKernelGGL((k_post_only<PostFn>), dim3(1), dim3(1), 0, stream, totals, post);
    return hipGetLastError();
  }
  hipLaunchKernelGGL((k_count_flags<NKEYS>), dim3(nb), dim3(kThreads), 0, stream, n, runs, flags, shift, counts);
  if ((size_t)NKEYS * runs > 16384)
    hipLaunchKernelGGL((k_scan<NKEYS, 1024>), dim3(1), dim3(1024), 0, stream, runs, counts, totals, 0u, (const int*)nullptr, 1u, (uint32_t)kFlagRun, n);
  else
    hipLaunchKernelGGL((k_scan<NKEYS, 256>), dim3(1), dim3(256), 0, stream, runs, counts, totals, 0u, (const int*)nullptr, 1u, (uint32_t)kFlagRun, n);
  hipLaunchKernelGGL((k_scatter_flags<NKEYS, EmitFn, PostFn>), dim3(nb), dim3(kThreads), 0, stream, n, runs, flags, shift, emit, counts,
                     totals, post);
  return hipGetLastError();
}

}  // namespace svc
