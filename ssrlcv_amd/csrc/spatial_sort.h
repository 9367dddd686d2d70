// ssrlcv_amd/csrc/spatial_sort.h -- internal interface of spatial_sort.hip (not part of the C ABI)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace svm {
// bytes of scratch sort_filled_keys needs for n elements (pure host arithmetic)
size_t sort_scratch_bytes(uint32_t n);
// the array inside `scratch` the caller fills (on `stream`) before sort_filled_keys: keys[n]
struct SortBuffers {
  uint32_t* keys;
};
SortBuffers sort_buffers(void* scratch, uint32_t n);
// perm[0..n) = 0 .. n-1 ordered by ascending (key, index): unique, so reproducible; asynchronous on `stream`
int sort_filled_keys(uint32_t n, uint32_t* perm, void* scratch, size_t scratchBytes, hipStream_t stream);
}  // namespace svm
