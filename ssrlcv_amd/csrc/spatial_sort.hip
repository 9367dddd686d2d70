// ssrlcv_amd/csrc/spatial_sort.hip -- spatial ordering of a feature set for the band-culled matcher (matcher.hip):
// perm[s] = index of the feature that comes s-th when features are ordered by (64-pixel row strip, x).  32 consecutive
// features of that order then cover a few pixels of x inside one strip, so a tile's bounding box is small and most
// (query tile, target tile) pairs can be rejected against the queries' epipolar bands without touching a descriptor.
// The sort itself is rocPRIM's device radix sort (a library sort, not a hot kernel); it lives in its own translation
// unit because the rocPRIM headers dominate the compile time.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include "ssrlcv_hip.h"
#include "spatial_sort.h"

namespace {
__global__ __launch_bounds__(256) void k_location_keys(const ssrlcv_sift_feature* __restrict__ feats, uint32_t n,
                                                       uint32_t* __restrict__ keys, uint32_t* __restrict__ iota) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  ssrlcv_float2 l = feats[i].loc;
  // NaN / negative coordinates clamp to 0, huge ones to the last strip: any order is correct, only culling suffers
  float x = l.x > 0.0f ? l.x : 0.0f, y = l.y > 0.0f ? l.y : 0.0f;
  uint32_t xi = x < 65535.0f ? (uint32_t)x : 65535u;
  uint32_t yb = y < 64.0f * 65535.0f ? (uint32_t)(y * (1.0f / 64.0f)) : 65535u;
  keys[i] = (yb << 16) | xi;
  iota[i] = i;
}
}  // namespace

namespace svm {

size_t sort_scratch_bytes(uint32_t n) {
  size_t nn = n ? n : 1;
  return 3 * ((nn * 4 + 255) / 256 * 256) + 16 * nn + (4u << 20);  // keys in/out, iota, rocPRIM temporary storage
}

int sort_by_location(const ssrlcv_sift_feature* feats, uint32_t n, uint32_t* perm, void* scratch, size_t scratchBytes,
                     hipStream_t stream) {
  if (n == 0) return SSRLCV_OK;
  if (scratchBytes < sort_scratch_bytes(n)) return SSRLCV_ERR_WORKSPACE;
  const size_t arr = ((size_t)n * 4 + 255) / 256 * 256;
  char* base = (char*)scratch;
  uint32_t* keysIn = (uint32_t*)base;
  uint32_t* keysOut = (uint32_t*)(base + arr);
  uint32_t* iota = (uint32_t*)(base + 2 * arr);
  void* tmp = base + 3 * arr;
  const size_t tmpAvail = scratchBytes - 3 * arr;
  hipLaunchKernelGGL(k_location_keys, dim3((n + 255) / 256), dim3(256), 0, stream, feats, n, keysIn, iota);
  size_t need = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, need, keysIn, keysOut, iota, perm, n, 0, 32, stream);
  if (e != hipSuccess) return (int)e;
  if (need > tmpAvail) return SSRLCV_ERR_WORKSPACE;
  e = rocprim::radix_sort_pairs(tmp, need, keysIn, keysOut, iota, perm, n, 0, 32, stream);
  if (e != hipSuccess) return (int)e;
  return SSRLCV_OK;
}

}  // namespace svm
