// ssrlcv_amd/csrc/spatial_sort.hip -- the sort behind the spatial orders of the band-culled matcher (matcher.hip): the caller
// fills 32-bit keys (a strip index of the set's frame in the high half, the position along the strip in the low half, see
// matcher.hip "band culling") and gets the permutation that orders them.  The sort itself is rocPRIM's device radix sort
// (a library sort, not a hot kernel); it lives in its own translation unit because the rocPRIM headers dominate the
// compile time.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include "ssrlcv_hip.h"
#include "spatial_sort.h"

namespace svm {

static size_t array_bytes(uint32_t n) { return ((size_t)(n ? n : 1) * 4 + 255) / 256 * 256; }

size_t sort_scratch_bytes(uint32_t n) {
  size_t nn = n ? n : 1;
  return 3 * array_bytes(n) + 16 * nn + (4u << 20);  // keys in/out, iota, rocPRIM temporary storage
}

SortBuffers sort_buffers(void* scratch, uint32_t n) {
  char* base = (char*)scratch;
  return SortBuffers{(uint32_t*)base, (uint32_t*)(base + 2 * array_bytes(n))};
}

int sort_filled_keys(uint32_t n, uint32_t* perm, void* scratch, size_t scratchBytes, hipStream_t stream) {
  if (n == 0) return SSRLCV_OK;
  if (scratchBytes < sort_scratch_bytes(n)) return SSRLCV_ERR_WORKSPACE;
  const size_t arr = array_bytes(n);
  char* base = (char*)scratch;
  uint32_t* keysIn = (uint32_t*)base;
  uint32_t* keysOut = (uint32_t*)(base + arr);
  uint32_t* iota = (uint32_t*)(base + 2 * arr);
  void* tmp = base + 3 * arr;
  const size_t tmpAvail = scratchBytes - 3 * arr;
  size_t need = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, need, keysIn, keysOut, iota, perm, n, 0, 32, stream);
  if (e != hipSuccess) return (int)e;
  if (need > tmpAvail) return SSRLCV_ERR_WORKSPACE;
  e = rocprim::radix_sort_pairs(tmp, need, keysIn, keysOut, iota, perm, n, 0, 32, stream);
  if (e != hipSuccess) return (int)e;
  return SSRLCV_OK;
}

}  // namespace svm
