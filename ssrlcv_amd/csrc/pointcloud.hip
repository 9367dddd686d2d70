// ssrlcv_amd/csrc/pointcloud.hip -- point-cloud leg of the hot path for gfx950.
//
// All kernels are per-match streaming kernels: a few dozen bytes in, 12-24 bytes out, bound by HBM/launch latency.
// One thread per bundle, 256-thread blocks (4 wave64), error sums reduced per wave with DPP/shuffle and one float
// atomic per wave (the reference uses a per-block shared atomicAdd, then a global atomicAdd: same non-deterministic
// float order, src/PointCloudFactory.cu:4533-4535).
#include <hip/hip_runtime.h>
#include "device_math.h"
#include "ssrlcv_hip.h"

using namespace sv;

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ ssrlcv_line make_line(const ssrlcv_camera& cam, ssrlcv_keypoint kp) {
  // src/PointCloudFactory.cu:4179-4195
  float dpix_x = (cam.foc * sv_tanf(cam.fov.x / 2.0f)) / (cam.size.x / 2.0f);
  float dpix_y = dpix_x;
  f3 v = mk3(dpix_x * ((kp.loc.x) - (cam.size.x / 2.0f)), dpix_y * ((kp.loc.y) - (cam.size.y / 2.0f)), cam.foc);
  v = rotate_point(v, cam.cam_rot);
  ssrlcv_line l;
  l.vec = normalize(v);
  l.pnt = cam.cam_pos;
  return l;
}

__global__ __launch_bounds__(kBlock) void k_generate_bundles(const ssrlcv_multimatch* __restrict__ matches,
                                                             const ssrlcv_keypoint* __restrict__ keyPoints,
                                                             uint32_t numBundles,
                                                             const ssrlcv_camera* __restrict__ cameras,
                                                             ssrlcv_bundle* __restrict__ bundles,
                                                             ssrlcv_line* __restrict__ lines) {
  uint32_t g = blockIdx.x * kBlock + threadIdx.x;
  if (g >= numBundles) return;
  ssrlcv_multimatch m = matches[g];
  ssrlcv_bundle b;
  b.numLines = m.numKeyPoints;
  b.index = m.index;
  b.invalid = 0;
  bundles[g] = b;
  int end = (int)m.numKeyPoints + m.index;
  for (int i = m.index; i < end; ++i) {
    ssrlcv_keypoint kp = keyPoints[i];
    lines[i] = make_line(cameras[kp.parentId], kp);
  }
}

// src/PointCloudFactory.cu:4201-4283
__global__ __launch_bounds__(kBlock) void k_generate_pushbroom_bundles(const ssrlcv_multimatch* __restrict__ matches,
                                                                       const ssrlcv_keypoint* __restrict__ keyPoints,
                                                                       uint32_t numBundles,
                                                                       const ssrlcv_pushbroom* __restrict__ pushbrooms,
                                                                       ssrlcv_bundle* __restrict__ bundles,
                                                                       ssrlcv_line* __restrict__ lines) {
  uint32_t g = blockIdx.x * kBlock + threadIdx.x;
  if (g >= numBundles) return;
  ssrlcv_multimatch m = matches[g];
  ssrlcv_bundle b;
  b.numLines = m.numKeyPoints;
  b.index = m.index;
  b.invalid = 0;
  bundles[g] = b;
  int end = (int)m.numKeyPoints + m.index;
  for (int i = m.index; i < end; ++i) {
    ssrlcv_keypoint kp = keyPoints[i];
    ssrlcv_pushbroom pb = pushbrooms[kp.parentId];
    float cx = (pb.size.x / 2.0f), cy = (pb.size.y / 2.0f);
    f3 k = mk3(pb.dpix.x * ((kp.loc.x) - cx), 0.0f, (-1.0f * pb.foc));
    float roll = (float)(pb.roll * (SSRLCV_PI_D / 180.0f));  // PI is a double macro upstream
    float radius = pb.axis_radius;
    float altitude = pb.altitude;
    float t = sv_tanf((float)(roll - (SSRLCV_PI_D / 2.0f)));
    float a = 1.0f + (t * t);
    float bq = -2.0f * radius * t;
    float c = radius * radius - ((altitude + radius) * (altitude + radius));
    float solution1 = (-1.0f * bq + sqrtf((bq * bq) - (4.0f * a * c))) / (2.0f * a);
    float solution2 = (-1.0f * bq - sqrtf((bq * bq) - (4.0f * a * c))) / (2.0f * a);
    f3 position;
    if (solution1 > 0) position = mk3(solution1, 0.0f, t * solution1 * -1.0f);
    else position = mk3(solution2, 0.0f, t * solution2 * -1.0f);
    float arc_length = (pb.gsd * (kp.loc.y - cy));
    float angle_out = arc_length / radius;
    k = rotate_point(k, mk3(0.0f, roll, 0.0f));
    position = rotate_point(position, mk3(angle_out, 0.0f, 0.0f));
    k.x = position.x - (k.x);
    k.y = position.y - (k.y);
    k.z = position.z - (k.z);
    ssrlcv_line l;
    l.vec = normalize(mk3(position.x - k.x, position.y - k.y, position.z - k.z));
    l.pnt = position;
    lines[i] = l;
  }
}

// Skew-line closest points (src/PointCloudFactory.cu:4506-4530).  Returns ||s1-s2||^2, writes the midpoint.
__device__ __forceinline__ float two_view_point(const ssrlcv_line& L1, const ssrlcv_line& L2, f3& point) {
  f3 n2 = cross(L2.vec, cross(L1.vec, L2.vec));
  f3 n1 = cross(L1.vec, cross(L1.vec, L2.vec));
  float numer1 = dot(sub(L2.pnt, L1.pnt), n2);
  float numer2 = dot(sub(L1.pnt, L2.pnt), n1);
  float denom1 = dot(L1.vec, n2);
  float denom2 = dot(L2.vec, n1);
  f3 s1 = add(L1.pnt, lscale(numer1 / denom1, L1.vec));
  f3 s2 = add(L2.pnt, lscale(numer2 / denom2, L2.vec));
  point = divs(add(s1, s2), 2.0f);
  // :4532, inside the kernel's own body upstream: fused by the rule of device_math.h
  const float dx = s1.x - s2.x, dy = s1.y - s2.y, dz = s1.z - s2.z;
  return __builtin_fmaf(dz, dz, nv_pp(dx, dx, dy, dy));
}

__global__ __launch_bounds__(kBlock) void k_triangulate2(const ssrlcv_line* __restrict__ lines,
                                                         ssrlcv_bundle* __restrict__ bundles, uint32_t n,
                                                         ssrlcv_float3* __restrict__ points, float* __restrict__ errors,
                                                         const float* __restrict__ cutoff, float* __restrict__ errorSum) {
  uint32_t g = blockIdx.x * kBlock + threadIdx.x;
  float error = 0.0f;
  if (g < n) {
    int idx = bundles[g].index;
    ssrlcv_line L1 = lines[idx];
    ssrlcv_line L2 = lines[idx + 1];
    f3 p;
    error = two_view_point(L1, L2, p);
    if (points) points[g] = p;
    if (errors) errors[g] = error;
    bundles[g].invalid = cutoff ? (uint8_t)(error > *cutoff) : (uint8_t)0;
  }
  if (errorSum) {
    float s = wave_sum(error);
    if ((threadIdx.x & 63) == 0) atomicAdd(errorSum, s);
  }
}

__global__ __launch_bounds__(kBlock) void k_triangulateN(const ssrlcv_line* __restrict__ lines,
                                                         ssrlcv_bundle* __restrict__ bundles, uint32_t n,
                                                         ssrlcv_float3* __restrict__ points, float* __restrict__ errors,
                                                         const float* __restrict__ cutoff, float* __restrict__ errorSum,
                                                         int noErrorVariant) {
  uint32_t g = blockIdx.x * kBlock + threadIdx.x;
  float a_error = 0.0f;
  if (g < n) {
    ssrlcv_bundle bd = bundles[g];
    f3 S[3] = {mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0)};
    f3 C = mk3(0, 0, 0);
    int lo = bd.index, hi = bd.index + (int)bd.numLines;
    for (int i = lo; i < hi; ++i) {
      ssrlcv_line L1 = lines[i];
      f3 v = normalize(L1.vec);
      f3 tmp[3];
      tmp[0] = mk3(v.x * v.x, v.x * v.y, v.x * v.z);
      tmp[1] = mk3(v.y * v.x, v.y * v.y, v.y * v.z);
      tmp[2] = mk3(v.z * v.x, v.z * v.y, v.z * v.z);
      tmp[0].x -= 1;
      tmp[1].y -= 1;
      tmp[2].z -= 1;
      S[0] = add(S[0], tmp[0]);
      S[1] = add(S[1], tmp[1]);
      S[2] = add(S[2], tmp[2]);
      C = add(C, mul33(tmp, L1.pnt));
    }
    f3 Inv[3];
    f3 point = mk3(0, 0, 0);
    bool ok = inverse3(S, Inv);
    if (ok) {
      point = mul33(Inv, C);
      if (points) points[g] = point;
    }
    if (noErrorVariant) {
      if (!ok) bundles[g].invalid = 1;  // :4923-4926
    } else {
      for (int i = lo; i < hi; ++i) {
        ssrlcv_line L = lines[i];
        f3 lp1 = L.pnt;
        f3 lp2 = add(L.pnt, scale(L.vec, 1000.0f));
        f3 a = sub(point, lp1);
        f3 b = sub(point, lp2);
        f3 c = sub(lp2, lp1);
        f3 d = cross(a, b);
        float numer = mag(d);
        float denom = mag(c);
        a_error = numer / denom;  // '=' upstream (:5001): last line wins
        a_error *= a_error;
      }
      a_error /= (float)bd.numLines;
      if (errors) errors[g] = a_error;
      if (cutoff) bundles[g].invalid = (uint8_t)(a_error > *cutoff);
    }
  }
  if (errorSum && !noErrorVariant) {
    float s = wave_sum(a_error);
    if ((threadIdx.x & 63) == 0) atomicAdd(errorSum, s);
  }
}

// ---- BA sweep: all K camera-parameter sets in one launch ------------------------------------------------------
// Each thread keeps its match (two key points) in registers and walks the K parameter sets; the per-set camera pair
// is wave-uniform (scalar loads).  Per set: one wave reduction + one atomic per wave.
__global__ __launch_bounds__(kBlock) void k_ba_sweep2(const ssrlcv_multimatch* __restrict__ matches,
                                                      const ssrlcv_keypoint* __restrict__ keyPoints, uint32_t numBundles,
                                                      const ssrlcv_camera* __restrict__ cameras, uint32_t numCameras,
                                                      const float* __restrict__ params, uint32_t K,
                                                      float* __restrict__ errorSums) {
  uint32_t g = blockIdx.x * kBlock + threadIdx.x;
  bool active = g < numBundles;
  ssrlcv_keypoint kp0, kp1;
  ssrlcv_camera c0, c1;
  if (active) {
    ssrlcv_multimatch m = matches[g];
    kp0 = keyPoints[m.index];
    kp1 = keyPoints[m.index + 1];
    c0 = cameras[kp0.parentId];
    c1 = cameras[kp1.parentId];
  }
  for (uint32_t k = blockIdx.y; k < K; k += gridDim.y) {
    float error = 0.0f;
    if (active) {
      const float* p0 = params + ((size_t)k * numCameras + kp0.parentId) * 6;
      const float* p1 = params + ((size_t)k * numCameras + kp1.parentId) * 6;
      c0.cam_pos = mk3(p0[0], p0[1], p0[2]);
      c0.cam_rot = mk3(p0[3], p0[4], p0[5]);
      c1.cam_pos = mk3(p1[0], p1[1], p1[2]);
      c1.cam_rot = mk3(p1[3], p1[4], p1[5]);
      ssrlcv_line L1 = make_line(c0, kp0);
      ssrlcv_line L2 = make_line(c1, kp1);
      f3 p;
      error = two_view_point(L1, L2, p);
    }
    float s = wave_sum(error);
    if ((threadIdx.x & 63) == 0) atomicAdd(&errorSums[k], s);
  }
}

inline unsigned blocks_for(uint32_t n) { return (n + kBlock - 1) / kBlock; }

}  // namespace

extern "C" {

int ssrlcv_hip_generate_bundles(const ssrlcv_multimatch* matches, const ssrlcv_keypoint* keyPoints, uint32_t numBundles,
                                const ssrlcv_camera* cameras, uint32_t numCameras, ssrlcv_bundle* bundles,
                                ssrlcv_line* lines, ssrlcv_stream_t stream) {
  if (!matches || !keyPoints || !cameras || !bundles || !lines || numCameras == 0) return SSRLCV_ERR_INVALID_ARG;
  if (numBundles == 0) return SSRLCV_OK;
  hipLaunchKernelGGL(k_generate_bundles, dim3(blocks_for(numBundles)), dim3(kBlock), 0, (hipStream_t)stream, matches,
                     keyPoints, numBundles, cameras, bundles, lines);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

int ssrlcv_hip_generate_pushbroom_bundles(const ssrlcv_multimatch* matches, const ssrlcv_keypoint* keyPoints,
                                          uint32_t numBundles, const ssrlcv_pushbroom* pushbrooms, uint32_t numCameras,
                                          ssrlcv_bundle* bundles, ssrlcv_line* lines, ssrlcv_stream_t stream) {
  if (!matches || !keyPoints || !pushbrooms || !bundles || !lines || numCameras == 0) return SSRLCV_ERR_INVALID_ARG;
  if (numBundles == 0) return SSRLCV_OK;
  hipLaunchKernelGGL(k_generate_pushbroom_bundles, dim3(blocks_for(numBundles)), dim3(kBlock), 0, (hipStream_t)stream,
                     matches, keyPoints, numBundles, pushbrooms, bundles, lines);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

int ssrlcv_hip_triangulate2(const ssrlcv_line* lines, ssrlcv_bundle* bundles, uint32_t numBundles, ssrlcv_float3* points,
                            float* errors, const float* cutoff, float* errorSum, ssrlcv_stream_t stream) {
  if (!lines || !bundles) return SSRLCV_ERR_INVALID_ARG;
  if (numBundles == 0) return SSRLCV_OK;
  hipLaunchKernelGGL(k_triangulate2, dim3(blocks_for(numBundles)), dim3(kBlock), 0, (hipStream_t)stream, lines, bundles,
                     numBundles, points, errors, cutoff, errorSum);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

int ssrlcv_hip_triangulateN(const ssrlcv_line* lines, ssrlcv_bundle* bundles, uint32_t numBundles, ssrlcv_float3* points,
                            float* errors, const float* cutoff, float* errorSum, int noErrorVariant,
                            ssrlcv_stream_t stream) {
  if (!lines || !bundles) return SSRLCV_ERR_INVALID_ARG;
  if (numBundles == 0) return SSRLCV_OK;
  hipLaunchKernelGGL(k_triangulateN, dim3(blocks_for(numBundles)), dim3(kBlock), 0, (hipStream_t)stream, lines, bundles,
                     numBundles, points, errors, cutoff, errorSum, noErrorVariant);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

size_t ssrlcv_hip_ba_sweep2_workspace_bytes(uint32_t, uint32_t) { return 0; }

int ssrlcv_hip_ba_sweep2(const ssrlcv_multimatch* matches, const ssrlcv_keypoint* keyPoints, uint32_t numBundles,
                         const ssrlcv_camera* cameras, uint32_t numCameras, const float* params, uint32_t K,
                         float* errorSums, void*, size_t, ssrlcv_stream_t stream) {
  if (!matches || !keyPoints || !cameras || !params || !errorSums || numCameras == 0) return SSRLCV_ERR_INVALID_ARG;
  if (K == 0) return SSRLCV_OK;
  SSRLCV_HIP_TRY(hipMemsetAsync(errorSums, 0, sizeof(float) * K, (hipStream_t)stream));
  if (numBundles == 0) return SSRLCV_OK;
  unsigned bx = blocks_for(numBundles);
  // enough blocks to fill 256 CUs several times over; each block walks K/gridDim.y parameter sets
  unsigned by = 1;
  while (bx * by < 2048 && by < K) by *= 2;
  if (by > K) by = K;
  hipLaunchKernelGGL(k_ba_sweep2, dim3(bx, by), dim3(kBlock), 0, (hipStream_t)stream, matches, keyPoints, numBundles,
                     cameras, numCameras, params, K, errorSums);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

}  // extern "C"
