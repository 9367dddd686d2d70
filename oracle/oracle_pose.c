/* oracle/oracle_pose.c -- TEST INFRASTRUCTURE ONLY (see oracle.h): CPU restatement of the device part of
 * ssrlcv::PoseEstimator::LM_iteration (reference src/PoseEstimator.cu).  parity unpinned: the reference holds no
 * test or fixture for the pose estimator; this file pins the HIP kernels to the
 * reference's formulas only.
 *
 *   getResidual                      src/PoseEstimator.cu:742-812   oracle_pose_residual
 *   computeResidualsAndJacobian      src/PoseEstimator.cu:647-729   (inside oracle_pose_lm_terms)
 *   computeJTJ / computeJTf          src/PoseEstimator.cu:814-844   (inside oracle_pose_lm_terms)
 *   computeCost                      src/PoseEstimator.cu:731-740   oracle_pose_cost
 *
 * Quirks kept: the residual is the gap between the two closest points of the rays (s1 - s2), its 4th component is 0;
 * only the three rotation columns of the Jacobian are finite differences (delta = 1e-5, central), the three position
 * columns are written as 0 although the perturbed residuals are evaluated; JTJ is accumulated as out[i + 6 j].  The
 * reference sums with float atomicAdd in arbitrary order; here the sums run in match order. */
#include "oracle.h"
#include "oracle_math.h"

static void residual(const o_pose* pose, const o_camera* query, const o_camera* target, o_float2 q_loc, o_float2 t_loc,
                     float out[4]) {
  o_float3 queryPnt = f3(0, 0, 0);
  o_float3 queryVec = f3(query->dpix.x * ((q_loc.x) - (query->size.x / 2.0f)),
                         query->dpix.y * ((q_loc.y) - (query->size.y / 2.0f)), query->foc);
  queryVec = f3_normalize(queryVec);
  o_float3 targetPnt = f3(pose->x, pose->y, pose->z);
  o_float3 targetVec = f3(target->dpix.x * ((t_loc.x) - (target->size.x / 2.0f)),
                          target->dpix.y * ((t_loc.y) - (target->size.y / 2.0f)), target->foc);
  targetVec = rotate_point(targetVec, f3(pose->roll, pose->pitch, pose->yaw));
  targetVec = f3_normalize(targetVec);
  /* closest points of the two rays (same construction as twoViewTriangulate) */
  o_float3 n2 = f3_cross(targetVec, f3_cross(queryVec, targetVec));
  o_float3 n1 = f3_cross(queryVec, f3_cross(queryVec, targetVec));
  float numer1 = f3_dot(f3_sub(targetPnt, queryPnt), n2);
  float numer2 = f3_dot(f3_sub(queryPnt, targetPnt), n1);
  float denom1 = f3_dot(queryVec, n2);
  float denom2 = f3_dot(targetVec, n1);
  o_float3 s1 = f3_add(queryPnt, f3_lscale(numer1 / denom1, queryVec));
  o_float3 s2 = f3_add(targetPnt, f3_lscale(numer2 / denom2, targetVec));
  out[0] = s1.x - s2.x;
  out[1] = s1.y - s2.y;
  out[2] = s1.z - s2.z;
  out[3] = 0;
}

void oracle_pose_residual(const o_pose* pose, const o_camera* query, const o_camera* target, const o_float2* q_loc,
                          const o_float2* t_loc, float out[4]) {
  residual(pose, query, target, *q_loc, *t_loc, out);
}

float oracle_pose_cost(const o_match* matches, uint32_t n, const o_pose* pose, const o_camera* query,
                       const o_camera* target) {
  float cost = 0;
  for (uint32_t m = 0; m < n; ++m) {
    float r[4];
    residual(pose, query, target, matches[m].keyPoints[0].loc, matches[m].keyPoints[1].loc, r);
    float sum = r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
    cost += sum;
  }
  return cost;
}

void oracle_pose_lm_terms(const o_match* matches, uint32_t n, const o_pose* pose0, const o_camera* query,
                          const o_camera* target, float JTJ[36], float JTf[6], float* cost) {
  for (int i = 0; i < 36; ++i) JTJ[i] = 0;
  for (int i = 0; i < 6; ++i) JTf[i] = 0;
  float csum = 0;
  for (uint32_t m = 0; m < n; ++m) {
    o_float2 q_loc = matches[m].keyPoints[0].loc, t_loc = matches[m].keyPoints[1].loc;
    o_pose pose = *pose0;
    float f[4], J[4][6];
    const float delta = 1e-5;
    residual(&pose, query, target, q_loc, t_loc, f);
    float* ang[3] = {&pose.roll, &pose.pitch, &pose.yaw};
    for (int c = 0; c < 3; ++c) {
      float right[4], left[4];
      float saved = *ang[c];
      *ang[c] += delta;
      residual(&pose, query, target, q_loc, t_loc, right);
      *ang[c] -= 2 * delta;
      residual(&pose, query, target, q_loc, t_loc, left);
      *ang[c] = saved;
      for (int r = 0; r < 4; ++r) J[r][c] = (right[r] - left[r]) / (2 * delta);
    }
    for (int c = 3; c < 6; ++c)
      for (int r = 0; r < 4; ++r) J[r][c] = 0;
    for (int r = 0; r < 4; ++r) {
      for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < 6; ++j) JTJ[i + 6 * j] += J[r][i] * J[r][j];
        JTf[i] += J[r][i] * f[r];
      }
    }
    csum += f[0] * f[0] + f[1] * f[1] + f[2] * f[2] + f[3] * f[3];
  }
  if (cost) *cost = csum;
}
