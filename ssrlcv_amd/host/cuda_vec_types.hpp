// ssrlcv_amd/host/cuda_vec_types.hpp -- the CUDA vector types the reference API is written in, as plain structs with
// CUDA's alignments.  Defined in the global namespace under the CUDA names on purpose: Unity<T>::checkpoint stores
// typeid(T).name() (include/Unity.cuh:929-947), and "6float3" must stay "6float3" for the reference's .uty files to
// load (HIP's own float3 is HIP_vector_type<float,3>, whose mangled name differs -- SURVEY.md section 7).
// This header must therefore not be mixed with <hip/hip_runtime.h> in one translation unit; the host mirror reaches
// the GPU only through the C ABI (include/ssrlcv_hip.h).
#pragma once
#include <cstdint>

struct alignas(8) float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) uint2 { unsigned int x, y; };
struct alignas(8) int2 { int x, y; };

// the operators the hot path uses (src/cuda_vec_util.cu:136-138,559-563,1228-1250,1585-1605)
inline bool operator==(const uint2& a, const uint2& b) { return a.x == b.x && a.y == b.y; }
inline bool operator<(const uint2& a, const uint2& b) {
  if (a == b) return false;
  else if (a.x == b.x) return a.y < b.y;
  else return a.x < b.x;
}
inline float3 operator+(const float3& a, const float3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float3 operator-(const float3& a, const float3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3& operator-=(float3& a, const float3& b) { a = a - b; return a; }
inline float3 operator*(const float3& a, const float& b) { return {a.x * b, a.y * b, a.z * b}; }
inline float3 operator/(const float3& a, const float& b) { return {a.x / b, a.y / b, a.z / b}; }
inline uint2 operator*(const uint2& a, const unsigned int& b) { return {a.x * b, a.y * b}; }
inline uint2 operator/(const uint2& a, const uint2& b) { return {a.x / b.x, a.y / b.y}; }
