// ssrlcv_amd/host/Feature.hpp -- Feature<D> and SIFT_Descriptor (include/Feature.cuh:31-94, src/Feature.cu:7-42).
#pragma once
#include <cfloat>
#include "cuda_vec_types.hpp"
#include "ssrlcv_types.h"

namespace ssrlcv {

template <typename D>
struct Feature {
  int parent;    ///< parent image ID
  float2 loc;    ///< location on parent image
  D descriptor;  ///< descriptor of feature
  Feature() : parent(-1), loc{-1.0f, -1.0f} {}
  Feature(float2 loc) : parent(-1), loc(loc) {}
  Feature(float2 loc, D descriptor) : parent(-1), loc(loc), descriptor(descriptor) {}
};

struct SIFT_Descriptor {
  float sigma;
  float theta;
  unsigned char values[128];
  SIFT_Descriptor() : sigma(0.0f), theta(0.0f) {}
  SIFT_Descriptor(float theta) : sigma(0.0f), theta(theta) {}
  SIFT_Descriptor(float theta, unsigned char v[128]) : sigma(0.0f), theta(theta) {
    for (int i = 0; i < 128; ++i) values[i] = v[i];
  }
  // squared L2 with early exit (src/Feature.cu:36-42); host-side helper for tests and the reference's comparators
  float distProtocol(const SIFT_Descriptor& b, const float& bestMatch = FLT_MAX) const {
    float dist = 0.0f;
    for (int i = 0; i < 128 && dist < bestMatch; ++i)
      dist += ((float)values[i] - b.values[i]) * ((float)values[i] - b.values[i]);
    return dist;
  }
};

static_assert(sizeof(Feature<SIFT_Descriptor>) == sizeof(ssrlcv_sift_feature), "Feature<SIFT_Descriptor> must be 152 B");

}  // namespace ssrlcv
