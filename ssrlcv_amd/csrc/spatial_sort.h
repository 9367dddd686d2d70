// ssrlcv_amd/csrc/spatial_sort.h -- internal interface of spatial_sort.hip (not part of the C ABI)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "ssrlcv_types.h"

namespace svm {
// bytes of scratch sort_by_location needs for n features (pure host arithmetic)
size_t sort_scratch_bytes(uint32_t n);
// perm[0..n): indices of `feats` ordered by (64-pixel row strip, x); asynchronous on `stream`
int sort_by_location(const ssrlcv_sift_feature* feats, uint32_t n, uint32_t* perm, void* scratch, size_t scratchBytes,
                     hipStream_t stream);
}  // namespace svm
