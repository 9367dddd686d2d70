// ssrlcv_amd/csrc/scan_lookback.h -- single-pass prefix sums for gfx950 (chained scan with decoupled look-back).
//
// One launch: a block takes the next TILE of the input (tile ids are handed out by an atomic counter, so a tile's
// predecessors are always running or done), reduces it, publishes the tile's aggregate, looks back over the
// descriptors of the tiles before it until it meets one that already knows its inclusive prefix, publishes its own, and
// goes on to produce its outputs with the exclusive prefix in hand.  The input is read once and nothing is written but
// the result -- against the count -> scan -> scatter form (compact.h) or a library scan followed by a consumer kernel,
// both of which read the input twice and launch two or three times.
//
// Descriptor = one 64-bit word per tile and sum: bits 63:62 state (0 not there yet, 1 aggregate of this tile alone,
// 2 inclusive prefix up to and including this tile), bits 31:0 the value.  State and value travel in ONE word, written
// and read with relaxed device-scope atomics, so no fence orders them.  NSUM sums are carried side by side (a stream
// compaction of variable-length records needs two: records kept, elements kept).
//
// The caller zeroes the workspace (workspace_bytes: descriptors + the tile counter) on the stream before the launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svs {

constexpr int kThreads = 256;
constexpr unsigned long long kAggregate = 1ull << 62, kPrefix = 2ull << 62, kStateMask = 3ull << 62;

template <int NSUM>
struct TileScan {
  unsigned long long* desc;  // [NSUM][numTiles]
  uint32_t* counter;         // next tile to hand out
  uint32_t numTiles;
};
template <int NSUM>
inline size_t workspace_bytes(uint32_t numTiles) {
  return ((size_t)NSUM * numTiles * 8 + 255) / 256 * 256 + 256;
}
template <int NSUM>
inline TileScan<NSUM> make_tile_scan(void* workspace, uint32_t numTiles) {
  TileScan<NSUM> t;
  t.desc = (unsigned long long*)workspace;
  t.counter = (uint32_t*)((char*)workspace + ((size_t)NSUM * numTiles * 8 + 255) / 256 * 256);
  t.numTiles = numTiles;
  return t;
}

// block-wide: the tile this block works on next (>= numTiles: none left)
__device__ __forceinline__ uint32_t next_tile(uint32_t* counter) {
  __shared__ uint32_t s_tile;
  __syncthreads();  // the previous tile's readers of s_tile are through
  if (threadIdx.x == 0) s_tile = atomicAdd(counter, 1u);
  __syncthreads();
  return s_tile;
}

// wave64 inclusive scan / sum
__device__ __forceinline__ uint32_t wave_inclusive(uint32_t v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(v, o, 64);
    if ((threadIdx.x & 63) >= (unsigned)o) v += y;
  }
  return v;
}
__device__ __forceinline__ uint32_t wave_total(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide exclusive scan of one value per thread (256 threads) -> this thread's exclusive offset inside the tile; the
// tile's total comes back in `total` (every thread).
template <int NSUM>
__device__ __forceinline__ void block_exclusive(const uint32_t (&mine)[NSUM], uint32_t (&excl)[NSUM], uint32_t (&total)[NSUM]) {
  __shared__ uint32_t s_wave[NSUM][kThreads / 64];
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl[NSUM];
#pragma unroll
  for (int s = 0; s < NSUM; ++s) {
    incl[s] = wave_inclusive(mine[s]);
    if (lane == 63) s_wave[s][wave] = incl[s];
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < NSUM; ++s) {
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) {
      const uint32_t v = s_wave[s][w];
      before += (unsigned)w < wave ? v : 0u;
      all += v;
    }
    excl[s] = before + incl[s] - mine[s];
    total[s] = all;
  }
  __syncthreads();  // s_wave may be reused by the next tile
}

// Block-wide: publish this tile's aggregates, look back, publish its inclusive prefixes -> the exclusive prefix of the
// tile (every thread).  Wave 0 does the look-back: lane l reads the descriptor of tile (tile - 1 - l - 64 k).
template <int NSUM>
__device__ __forceinline__ void tile_prefix(const TileScan<NSUM>& ts, uint32_t tile, const uint32_t (&aggregate)[NSUM],
                                            uint32_t (&exclusive)[NSUM]) {
  __shared__ uint32_t s_excl[NSUM];
  if (threadIdx.x < 64) {
    const unsigned lane = threadIdx.x;
    if (lane == 0) {
#pragma unroll
      for (int s = 0; s < NSUM; ++s)
        __hip_atomic_store(&ts.desc[(size_t)s * ts.numTiles + tile], (tile == 0 ? kPrefix : kAggregate) | aggregate[s], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    uint32_t run[NSUM];
#pragma unroll
    for (int s = 0; s < NSUM; ++s) run[s] = 0;
    if (tile != 0) {
      for (int s = 0; s < NSUM; ++s) {
        uint32_t sum = 0;
        int64_t at = (int64_t)tile - 1 - lane;  // this lane's predecessor in the current window
        while (true) {
          unsigned long long d = kPrefix;  // lanes past tile 0 read as "prefix 0": they end the search without adding
          if (at >= 0) {
            do {
              d = __hip_atomic_load(&ts.desc[(size_t)s * ts.numTiles + (size_t)at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } while ((d & kStateMask) == 0ull);
          }
          const unsigned long long isPrefix = __ballot((d & kStateMask) == kPrefix);
          // descriptors from the nearest predecessor (lane 0) up to and including the first prefix count
          const int stop = __ffsll((long long)isPrefix) - 1;  // >= 0 unless the whole window is aggregates
          const bool counts = stop < 0 || (int)lane <= stop;
          sum += wave_total(counts ? (uint32_t)d : 0u);
          if (stop >= 0) break;
          at -= 64;
        }
        run[s] = sum;
      }
      if (lane == 0) {
#pragma unroll
        for (int s = 0; s < NSUM; ++s)
          __hip_atomic_store(&ts.desc[(size_t)s * ts.numTiles + tile], kPrefix | (unsigned long long)(uint32_t)(run[s] + aggregate[s]),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int s = 0; s < NSUM; ++s) s_excl[s] = run[s];
    }
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < NSUM; ++s) exclusive[s] = s_excl[s];
}

// ---- plain exclusive scan of a uint32 array (out[i] = sum of in[0..i)), ITEMS per thread; in == out is fine
template <int ITEMS>
__global__ __launch_bounds__(kThreads) void k_exclusive_scan(const uint32_t* in, uint32_t* out, uint32_t n, TileScan<1> ts) {
  constexpr uint32_t kTile = kThreads * ITEMS;
  for (uint32_t tile = next_tile(ts.counter); tile < ts.numTiles; tile = next_tile(ts.counter)) {
    const uint32_t base = tile * kTile + threadIdx.x * ITEMS;
    uint32_t v[ITEMS], mine[1] = {0};
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      v[j] = base + j < n ? in[base + j] : 0u;
      mine[0] += v[j];
    }
    uint32_t excl[1], total[1], prefix[1];
    block_exclusive<1>(mine, excl, total);
    tile_prefix<1>(ts, tile, total, prefix);
    uint32_t run = prefix[0] + excl[0];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      if (base + j < n) out[base + j] = run;
      run += v[j];
    }
  }
}
// queues memset + the scan; workspace of workspace_bytes<1>(tiles(n)) bytes
template <int ITEMS = 8>
inline uint32_t scan_tiles(uint32_t n) { return (n + kThreads * ITEMS - 1) / (kThreads * ITEMS); }
template <int ITEMS = 8>
inline hipError_t exclusive_scan(const uint32_t* in, uint32_t* out, uint32_t n, void* workspace, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const uint32_t tiles = scan_tiles<ITEMS>(n);
  hipError_t e = hipMemsetAsync(workspace, 0, workspace_bytes<1>(tiles), st);
  if (e != hipSuccess) return e;
  const TileScan<1> ts = make_tile_scan<1>(workspace, tiles);
  const unsigned blocks = tiles < 2048u ? tiles : 2048u;
  hipLaunchKernelGGL((k_exclusive_scan<ITEMS>), dim3(blocks), dim3(kThreads), 0, st, in, out, n, ts);
  return hipGetLastError();
}

}  // namespace svs
