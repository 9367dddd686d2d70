// ssrlcv_amd/csrc/spatial_sort.hip -- the sort behind the spatial orders of the band-culled matcher (matcher.hip): the caller
// fills 32-bit keys (a strip index of the pair's frame in the high half, the position along the strip in the low half, see
// matcher.hip "band culling") and gets the permutation that orders them.
//
// Round 4: hand-written for what the keys are (rounds 2-4 called rocPRIM's radix sort: nine launches and ~125 us for 4e5
// keys, twelve sorts per match stage of the 4-view flow = 1.5 of its 8.9 ms).  A key's strip takes a few hundred distinct
// values and a strip holds a few thousand keys, so: bucket by strip (4096 bins, (strip + 2048) mod 4096: contiguous for
// frame coordinates within +-32 768 px; beyond, strips alias into a bin and are separated again by the sort inside it),
// then sort every bin on its own in LDS.  Four kernels:
//   k_bin_hist     per-block LDS histogram of a contiguous chunk, non-empty bins flushed by one atomic each
//   k_bin_scan     exclusive scan of the 4096 counts (one block)
//   k_bin_scatter  the same chunks again: a block reserves its share of every bin with one atomic, ranks inside by LDS
//                  atomics, writes (key << 32 | index) -- the order inside a bin is arbitrary here ...
//   k_bin_sort     ... and total after this one: bitonic sort of the bin's 64-bit words in LDS (<= 4096 of them; a larger
//                  bin -- every feature on one line -- is sorted in place in global memory by the same network, slowly
//                  and correctly), perm = the low words.  The index is part of the sorted word, so the permutation is
//                  unique: equal keys come in index order, run to run.
#include <hip/hip_runtime.h>
#include "ssrlcv_hip.h"
#include "spatial_sort.h"

namespace {
constexpr int kBins = 4096;
constexpr int kChunkBlocks = 128;   // blocks of the two chunked passes
constexpr int kSortCap = 4096;      // 64-bit words of a bin sorted in LDS (32 KB)

__device__ __forceinline__ uint32_t bin_of(uint32_t key) { return ((key >> 16) + 2048u) & (uint32_t)(kBins - 1); }

__global__ __launch_bounds__(256) void k_bin_hist(const uint32_t* __restrict__ keys, uint32_t n, uint32_t chunk,
                                                  uint32_t* __restrict__ binCount) {
  __shared__ uint32_t s_h[kBins];
  for (int i = threadIdx.x; i < kBins; i += 256) s_h[i] = 0;
  __syncthreads();
  const uint32_t b0 = blockIdx.x * chunk, b1 = b0 + chunk < n ? b0 + chunk : n;
  for (uint32_t i = b0 + threadIdx.x; i < b1; i += 256) atomicAdd(&s_h[bin_of(keys[i])], 1u);
  __syncthreads();
  for (int i = threadIdx.x; i < kBins; i += 256)
    if (s_h[i]) atomicAdd(&binCount[i], s_h[i]);
}

// binBase[0 .. kBins] = exclusive scan of binCount; cursor[] = 0
__global__ __launch_bounds__(1024) void k_bin_scan(const uint32_t* __restrict__ binCount, uint32_t* __restrict__ binBase,
                                                   uint32_t* __restrict__ cursor) {
  __shared__ uint32_t s_w[16];
  const unsigned t = threadIdx.x, lane = t & 63, wave = t >> 6;
  uint32_t c[4], sum = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) { c[j] = binCount[4 * t + j]; sum += c[j]; }
  uint32_t incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = __shfl_up(incl, o, 64);
    if (lane >= (unsigned)o) incl += v;
  }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (unsigned w = 0; w < wave; ++w) base += s_w[w];
  uint32_t run = base + incl - sum;
#pragma unroll
  for (int j = 0; j < 4; ++j) { binBase[4 * t + j] = run; cursor[4 * t + j] = 0; run += c[j]; }
  if (t == 1023) binBase[kBins] = run;
}

__global__ __launch_bounds__(256) void k_bin_scatter(const uint32_t* __restrict__ keys, uint32_t n, uint32_t chunk,
                                                     const uint32_t* __restrict__ binBase, uint32_t* __restrict__ cursor,
                                                     unsigned long long* __restrict__ words) {
  __shared__ uint32_t s_h[kBins];   // this block's count per bin, then its running rank
  __shared__ uint32_t s_at[kBins];  // where this block's share of the bin starts
  for (int i = threadIdx.x; i < kBins; i += 256) s_h[i] = 0;
  __syncthreads();
  const uint32_t b0 = blockIdx.x * chunk, b1 = b0 + chunk < n ? b0 + chunk : n;
  for (uint32_t i = b0 + threadIdx.x; i < b1; i += 256) atomicAdd(&s_h[bin_of(keys[i])], 1u);
  __syncthreads();
  for (int i = threadIdx.x; i < kBins; i += 256) {
    const uint32_t c = s_h[i];
    s_at[i] = c ? binBase[i] + atomicAdd(&cursor[i], c) : 0u;
    s_h[i] = 0;
  }
  __syncthreads();
  for (uint32_t i = b0 + threadIdx.x; i < b1; i += 256) {
    const uint32_t k = keys[i], b = bin_of(k);
    const uint32_t pos = s_at[b] + atomicAdd(&s_h[b], 1u);
    words[pos] = ((unsigned long long)k << 32) | (unsigned long long)i;
  }
}

// Bitonic sort of w[0 .. cnt) in the all-ascending ("flip") formulation: for every block size k the first stage compares
// element i of a block with its mirror image k - 1 - i (which merges the two ascending halves without reversing one), the
// following stages compare i with i + j, j = k / 4 .. 1; every comparator puts the smaller word at the lower index.  With
// that, positions >= cnt behave as +inf without existing: a comparator that would touch one is a no-op, so any cnt works
// on the network of the next power of two, in LDS and in global memory alike.
template <class Ptr>
__device__ __forceinline__ void bitonic_ascending(Ptr w, uint32_t cnt) {
  uint32_t n2 = 1;
  while (n2 < cnt) n2 <<= 1;
  for (uint32_t k = 2; k <= n2; k <<= 1) {
    const uint32_t hk = k >> 1;
    for (uint32_t t = threadIdx.x; t < n2 / 2; t += blockDim.x) {
      const uint32_t blk = t / hk, off = t - blk * hk;
      const uint32_t i = blk * k + off, x = blk * k + (k - 1 - off);
      if (x < cnt) {
        const unsigned long long a = w[i], b = w[x];
        if (a > b) { w[i] = b; w[x] = a; }
      }
    }
    __syncthreads();
    for (uint32_t j = k >> 2; j > 0; j >>= 1) {
      for (uint32_t t = threadIdx.x; t < n2 / 2; t += blockDim.x) {
        const uint32_t i = 2 * t - (t & (j - 1)), x = i + j;
        if (x < cnt) {
          const unsigned long long a = w[i], b = w[x];
          if (a > b) { w[i] = b; w[x] = a; }
        }
      }
      __syncthreads();
    }
  }
}

constexpr int kSortThreads = 1024;  // a bin's network is ~70 dependent stages of a few pairs per thread: latency, so many waves
__global__ __launch_bounds__(kSortThreads) void k_bin_sort(const uint32_t* __restrict__ binBase, unsigned long long* __restrict__ words,
                                                  uint32_t* __restrict__ perm) {
  __shared__ unsigned long long s_w[kSortCap];
  const uint32_t b0 = binBase[blockIdx.x], cnt = binBase[blockIdx.x + 1] - b0;  // (block-uniform)
  if (cnt == 0) return;
  if (cnt <= (uint32_t)kSortCap) {
    for (uint32_t i = threadIdx.x; i < cnt; i += kSortThreads) s_w[i] = words[b0 + i];
    __syncthreads();
    bitonic_ascending(s_w, cnt);
    for (uint32_t i = threadIdx.x; i < cnt; i += kSortThreads) perm[b0 + i] = (uint32_t)s_w[i];
  } else {  // a bin beyond the LDS capacity (every feature in one strip ...): in place in global memory, rare and merely right
    unsigned long long* w = words + b0;
    bitonic_ascending(w, cnt);
    for (uint32_t i = threadIdx.x; i < cnt; i += kSortThreads) perm[b0 + i] = (uint32_t)w[i];
  }
}
}  // namespace

namespace svm {

static size_t round256(size_t b) { return (b + 255) / 256 * 256; }

size_t sort_scratch_bytes(uint32_t n) {
  const size_t nn = n ? n : 1;
  return round256(nn * 4) + round256(nn * 8) + 3 * round256((size_t)(kBins + 1) * 4);  // keys, words, counts / bases / cursors
}

SortBuffers sort_buffers(void* scratch, uint32_t n) {
  (void)n;
  return SortBuffers{(uint32_t*)scratch};
}

int sort_filled_keys(uint32_t n, uint32_t* perm, void* scratch, size_t scratchBytes, hipStream_t stream) {
  if (n == 0) return SSRLCV_OK;
  if (scratchBytes < sort_scratch_bytes(n)) return SSRLCV_ERR_WORKSPACE;
  char* base = (char*)scratch;
  const uint32_t* keys = (const uint32_t*)base;
  unsigned long long* words = (unsigned long long*)(base + round256((size_t)n * 4));
  uint32_t* binCount = (uint32_t*)((char*)words + round256((size_t)n * 8));
  uint32_t* binBase = (uint32_t*)((char*)binCount + round256((size_t)(kBins + 1) * 4));
  uint32_t* cursor = (uint32_t*)((char*)binBase + round256((size_t)(kBins + 1) * 4));
  hipError_t e = hipMemsetAsync(binCount, 0, (size_t)kBins * 4, stream);
  if (e != hipSuccess) return (int)e;
  const uint32_t chunk = (n + kChunkBlocks - 1) / kChunkBlocks;
  hipLaunchKernelGGL(k_bin_hist, dim3(kChunkBlocks), dim3(256), 0, stream, keys, n, chunk, binCount);
  hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, stream, (const uint32_t*)binCount, binBase, cursor);
  hipLaunchKernelGGL(k_bin_scatter, dim3(kChunkBlocks), dim3(256), 0, stream, keys, n, chunk, (const uint32_t*)binBase, cursor,
                     words);
  hipLaunchKernelGGL(k_bin_sort, dim3(kBins), dim3(kSortThreads), 0, stream, (const uint32_t*)binBase, words, perm);
  e = hipGetLastError();
  return e == hipSuccess ? SSRLCV_OK : (int)e;
}

}  // namespace svm

extern "C" {

size_t ssrlcv_hip_sort_workspace_bytes(uint32_t n) { return svm::sort_scratch_bytes(n); }

int ssrlcv_hip_sort_keys_u32(const uint32_t* keys, uint32_t n, uint32_t* perm, void* workspace, size_t workspaceBytes,
                             ssrlcv_stream_t stream) {
  if (n == 0) return SSRLCV_OK;
  if (!keys || !perm || !workspace) return SSRLCV_ERR_INVALID_ARG;
  if (workspaceBytes < svm::sort_scratch_bytes(n)) return SSRLCV_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const svm::SortBuffers sb = svm::sort_buffers(workspace, n);
  hipError_t e = hipMemcpyAsync(sb.keys, keys, (size_t)n * 4, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return (int)e;
  return svm::sort_filled_keys(n, perm, workspace, workspaceBytes, st);
}

}  // extern "C"
