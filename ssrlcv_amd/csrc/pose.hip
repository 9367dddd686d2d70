// ssrlcv_amd/csrc/pose.hip -- the device part of ssrlcv::PoseEstimator::LM_iteration (src/PoseEstimator.cu:349-393) as
// one fused kernel.  The reference launches computeResidualsAndJacobian (writes f[4M] and J[24M]), computeCost,
// computeJTJ (36 float atomics per Jacobian row) and computeJTf (6 per row); here every thread keeps its match's
// residual and the three non-zero Jacobian columns in registers, and only the 6 + 3 + 1 distinct sums leave the wave:
// no f / J arrays in HBM (96 + 16 B per match upstream), 10 atomics per wave instead of 168 per match.
// Algorithmic bytes: 40 B per match read (ssrlcv_match), 172 B written per launch.
#include <hip/hip_runtime.h>
#include "device_math.h"
#include "ssrlcv_hip.h"

namespace {
using sv::f3;

struct PoseArgs {
  ssrlcv_pose pose;
  ssrlcv_camera query, target;
};

// getResidual (src/PoseEstimator.cu:742-812): gap between the closest points of the query ray (camera frame of the
// query, origin) and the target ray (rotated by the relative pose, from pose.xyz).  The projections the reference
// also evaluates never reach its return value.
__device__ __forceinline__ void pose_residual(const ssrlcv_pose& pose, const ssrlcv_camera& query,
                                              const ssrlcv_camera& target, ssrlcv_float2 q_loc, ssrlcv_float2 t_loc,
                                              float (&out)[3]) {
  const f3 queryPnt = sv::mk3(0.0f, 0.0f, 0.0f);
  f3 queryVec = sv::mk3(query.dpix.x * ((q_loc.x) - (query.size.x / 2.0f)),
                        query.dpix.y * ((q_loc.y) - (query.size.y / 2.0f)), query.foc);
  queryVec = sv::normalize(queryVec);
  const f3 targetPnt = sv::mk3(pose.x, pose.y, pose.z);
  f3 targetVec = sv::mk3(target.dpix.x * ((t_loc.x) - (target.size.x / 2.0f)),
                         target.dpix.y * ((t_loc.y) - (target.size.y / 2.0f)), target.foc);
  targetVec = sv::rotate_point(targetVec, sv::mk3(pose.roll, pose.pitch, pose.yaw));
  targetVec = sv::normalize(targetVec);
  const f3 n2 = sv::cross(targetVec, sv::cross(queryVec, targetVec));
  const f3 n1 = sv::cross(queryVec, sv::cross(queryVec, targetVec));
  const float numer1 = sv::dot(sv::sub(targetPnt, queryPnt), n2);
  const float numer2 = sv::dot(sv::sub(queryPnt, targetPnt), n1);
  const float denom1 = sv::dot(queryVec, n2);
  const float denom2 = sv::dot(targetVec, n1);
  const f3 s1 = sv::add(queryPnt, sv::lscale(numer1 / denom1, queryVec));
  const f3 s2 = sv::add(targetPnt, sv::lscale(numer2 / denom2, targetVec));
  out[0] = s1.x - s2.x;
  out[1] = s1.y - s2.y;
  out[2] = s1.z - s2.z;
}

// out: JTJ[36] (JTJ[i + 6 j]), JTf[6], cost -- zeroed by the caller
__global__ __launch_bounds__(256) void k_pose_terms(const ssrlcv_match* __restrict__ matches, uint32_t n, PoseArgs a,
                                                    float* __restrict__ out) {
  float jtj[6] = {0, 0, 0, 0, 0, 0};  // (0,0) (0,1) (0,2) (1,1) (1,2) (2,2) of the rotation block
  float jtf[3] = {0, 0, 0};
  float cost = 0.0f;
  for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < n; m += gridDim.x * blockDim.x) {
    const ssrlcv_float2 q_loc = matches[m].keyPoints[0].loc, t_loc = matches[m].keyPoints[1].loc;
    ssrlcv_pose pose = a.pose;
    const float delta = 1e-5;
    float f[3], J[3][3];  // J[row][column]; the 4th row and the position columns are 0 (:698-727)
    pose_residual(pose, a.query, a.target, q_loc, t_loc, f);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float& angle = c == 0 ? pose.roll : c == 1 ? pose.pitch : pose.yaw;
      float right[3], left[3];
      const float saved = angle;
      angle += delta;
      pose_residual(pose, a.query, a.target, q_loc, t_loc, right);
      angle -= 2 * delta;
      pose_residual(pose, a.query, a.target, q_loc, t_loc, left);
      angle = saved;
#pragma unroll
      for (int r = 0; r < 3; ++r) J[r][c] = (right[r] - left[r]) / (2 * delta);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      jtj[0] += J[r][0] * J[r][0];
      jtj[1] += J[r][0] * J[r][1];
      jtj[2] += J[r][0] * J[r][2];
      jtj[3] += J[r][1] * J[r][1];
      jtj[4] += J[r][1] * J[r][2];
      jtj[5] += J[r][2] * J[r][2];
      jtf[0] += J[r][0] * f[r];
      jtf[1] += J[r][1] * f[r];
      jtf[2] += J[r][2] * f[r];
    }
    cost += f[0] * f[0] + f[1] * f[1] + f[2] * f[2];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) jtj[i] = sv::wave_sum(jtj[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) jtf[i] = sv::wave_sum(jtf[i]);
  cost = sv::wave_sum(cost);
  if ((threadIdx.x & 63) == 0) {
    const int ii[6] = {0, 0, 0, 1, 1, 2}, jj[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
    for (int k = 0; k < 6; ++k) atomicAdd(out + ii[k] + 6 * jj[k], jtj[k]);  // one triangle; mirrored afterwards
#pragma unroll
    for (int i = 0; i < 3; ++i) atomicAdd(out + 36 + i, jtf[i]);
    atomicAdd(out + 42, cost);
  }
}

// the other triangle of JTJ: copied, so the matrix is exactly symmetric whatever order the atomics arrived in
__global__ void k_pose_mirror(float* __restrict__ out) {
  const int i = threadIdx.x / 6, j = threadIdx.x % 6;
  if (i < j) out[j + 6 * i] = out[i + 6 * j];
}

// computeCost (src/PoseEstimator.cu:731-740)
__global__ __launch_bounds__(256) void k_pose_cost(const ssrlcv_match* __restrict__ matches, uint32_t n, PoseArgs a,
                                                   float* __restrict__ cost_out) {
  float cost = 0.0f;
  for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < n; m += gridDim.x * blockDim.x) {
    float f[3];
    pose_residual(a.pose, a.query, a.target, matches[m].keyPoints[0].loc, matches[m].keyPoints[1].loc, f);
    cost += f[0] * f[0] + f[1] * f[1] + f[2] * f[2];
  }
  cost = sv::wave_sum(cost);
  if ((threadIdx.x & 63) == 0) atomicAdd(cost_out, cost);
}

inline unsigned pose_blocks(uint32_t n) {
  unsigned b = (n + 255) / 256;
  return b < 1 ? 1 : b > 1024 ? 1024 : b;
}
}  // namespace

extern "C" {

int ssrlcv_hip_pose_lm_terms(const ssrlcv_match* matches, uint32_t numMatches, const ssrlcv_pose* pose,
                             const ssrlcv_camera* query, const ssrlcv_camera* target, float* out43,
                             ssrlcv_stream_t stream) {
  if (!matches || !pose || !query || !target || !out43) return SSRLCV_ERR_INVALID_ARG;
  PoseArgs a;
  a.pose = *pose;
  a.query = *query;
  a.target = *target;
  SSRLCV_HIP_TRY(hipMemsetAsync(out43, 0, 43 * sizeof(float), (hipStream_t)stream));
  if (numMatches == 0) return SSRLCV_OK;
  hipLaunchKernelGGL(k_pose_terms, dim3(pose_blocks(numMatches)), dim3(256), 0, (hipStream_t)stream, matches, numMatches, a,
                     out43);
  hipLaunchKernelGGL(k_pose_mirror, dim3(1), dim3(36), 0, (hipStream_t)stream, out43);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

int ssrlcv_hip_pose_cost(const ssrlcv_match* matches, uint32_t numMatches, const ssrlcv_pose* pose,
                         const ssrlcv_camera* query, const ssrlcv_camera* target, float* cost, ssrlcv_stream_t stream) {
  if (!matches || !pose || !query || !target || !cost) return SSRLCV_ERR_INVALID_ARG;
  PoseArgs a;
  a.pose = *pose;
  a.query = *query;
  a.target = *target;
  SSRLCV_HIP_TRY(hipMemsetAsync(cost, 0, sizeof(float), (hipStream_t)stream));
  if (numMatches == 0) return SSRLCV_OK;
  hipLaunchKernelGGL(k_pose_cost, dim3(pose_blocks(numMatches)), dim3(256), 0, (hipStream_t)stream, matches, numMatches, a,
                     cost);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

}  // extern "C"
