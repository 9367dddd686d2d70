// ssrlcv_amd/host/PoseEstimator.hpp -- ssrlcv::PoseEstimator's Levenberg-Marquardt refinement of the relative pose
// (include/PoseEstimator.cuh:30-61, src/PoseEstimator.cu:314-515) over the fused HIP terms kernel
// (ssrlcv_hip_pose_lm_terms / ssrlcv_hip_pose_cost).  estimatePoseRANSAC (7-point F-matrix RANSAC, :92-312) is not
// built: the reference's own stage flow has it commented out (src/Pipeline.cu:101) and starts LM from the
// cameras' relative pose.
//
// LM_iteration follows upstream step for step: terms at the current pose; then up to 20 trial steps with
// JTJ + lambda I, pseudo-inverse with singular values <= 1e-4 dropped (cusolverDnSgesvd upstream, pseudoInverse()
// here: PARITY UNPINNED), delta = -JTJ^+ JTf, trial cost, lambda doubled per trial; on success lambda /= 4.
#pragma once
#include <cstdio>
#include "Image.hpp"
#include "MatchFactory.hpp"
#include "matrix_util.hpp"

namespace ssrlcv {

struct FMatrixInliers {
  float fmatrix[3][3];
  unsigned long inliers;
  bool valid;
};

struct Pose {
  float roll;   // x-rotation, in radians
  float pitch;  // y-rotation, in radians
  float yaw;    // z-rotation, in radians
  float x;      // x-position, kilometers
  float y;      // y-position, kilometers
  float z;      // z-position, kilometers
};
static_assert(sizeof(Pose) == sizeof(ssrlcv_pose), "Pose layout");

class PoseEstimator {
 private:
  ptr::value<Unity<Match>> matches;  // keypoints should be (query, target)
  ptr::value<Image> query;
  ptr::value<Image> target;

  float cost(const Pose& pose) {
    ptr::device<float> c(1);
    HipSafeCall(ssrlcv_hip_pose_cost((const ssrlcv_match*)matches->device.get(), (uint32_t)matches->size(),
                                     (const ssrlcv_pose*)&pose, (const ssrlcv_camera*)&query->camera,
                                     (const ssrlcv_camera*)&target->camera, c.get(), nullptr));
    float h = 0;
    HipSafeCall(ssrlcv_hip_memcpy(&h, c.get(), sizeof(float), 1));
    return h;
  }

 public:
  // Sets up pose estimator to adjust target image
  PoseEstimator(ptr::value<Image> query, ptr::value<Image> target, ptr::value<Unity<Match>> matches)
      : matches(matches), query(query), target(target) {}

  Pose estimatePoseRANSAC() {
    logger.err << "PoseEstimator::estimatePoseRANSAC is not part of this build (SURVEY.md section 8f item 3)";
    exit(-1);
  }

  // PoseEstimator::LM_optimize (src/PoseEstimator.cu:314-341): the start position is the baseline between the two cameras
  // expressed in the query camera's frame (its z, y, x rotations undone in that order), in the matcher's length unit
  // (1 / 1000 of the cameras'); then at most kMaxLMIterations damped steps, stopping at the first one that does not
  // improve the cost.  The damping starts at 100, as upstream.
  static constexpr int kMaxLMIterations = 50;
  float3 baselineInQueryFrame() const {
    const float3 axis[3] = {{0, 0, 1}, {0, 1, 0}, {1, 0, 0}};
    const float angle[3] = {query->camera.cam_rot.z, query->camera.cam_rot.y, query->camera.cam_rot.x};
    float3 b = target->camera.cam_pos - query->camera.cam_pos;
    for (int k = 0; k < 3; ++k) b = rotatePointArbitrary(b, axis[k], -angle[k]);
    return b;
  }
  void LM_optimize(Pose* pose) {
    const float3 start = baselineInQueryFrame();
    pose->x = start.x / 1000;
    pose->y = start.y / 1000;
    pose->z = start.z / 1000;
    float damping = 100;
    for (int step = 1; step <= kMaxLMIterations; ++step) {
      logger.info.printf("Pose rotations: %f %f %f", pose->roll, pose->pitch, pose->yaw);
      logger.info.printf("Pose positions: %f %f %f", pose->x, pose->y, pose->z);
      if (!LM_iteration(pose, &damping)) break;
    }
  }

  // public here (private upstream) so that tests can drive single iterations
  bool LM_iteration(Pose* pose, float* lambda) {
    MemoryState origin = matches->getMemoryState();
    if (origin == cpu || matches->getFore() == cpu) matches->transferMemoryTo(gpu);
    ptr::device<float> terms(43);
    HipSafeCall(ssrlcv_hip_pose_lm_terms((const ssrlcv_match*)matches->device.get(), (uint32_t)matches->size(),
                                         (const ssrlcv_pose*)pose, (const ssrlcv_camera*)&query->camera,
                                         (const ssrlcv_camera*)&target->camera, terms.get(), nullptr));
    float h[43];
    HipSafeCall(ssrlcv_hip_memcpy(h, terms.get(), sizeof h, 1));
    float* JTJ = h;
    const float* JTf = h + 36;
    const float startCost = h[42];
    logger.info.printf("Starting cost: %f", startCost);
    float newCost = startCost + 100;  // just to make sure it starts off greater
    float old_lambda = 0;
    Pose newPose = *pose;
    int num_iterations = 0;
    bool ok = true;
    while (startCost <= newCost) {
      if (num_iterations >= 20) { ok = false; break; }
      num_iterations += 1;
      for (int i = 0; i < 6; i++) JTJ[i + 6 * i] += (*lambda - old_lambda);
      std::vector<float> JTJ_inv = pseudoInverse(JTJ, 6, 1e-4);
      float delta[6];
      for (int i = 0; i < 6; i++) {
        delta[i] = 0;
        for (int j = 0; j < 6; j++) delta[i] += -JTJ_inv[i * 6 + j] * JTf[j];
      }
      newPose.roll = pose->roll + delta[0];
      newPose.pitch = pose->pitch + delta[1];
      newPose.yaw = pose->yaw + delta[2];
      newPose.x = pose->x + delta[3];
      newPose.y = pose->y + delta[4];
      newPose.z = pose->z + delta[5];
      newCost = cost(newPose);
      logger.info.printf("New cost: %f", newCost);
      old_lambda = *lambda;
      *lambda *= 2;
    }
    if (ok) {
      *lambda /= 4;  // the loop's last doubling plus one real halving
      *pose = newPose;
    }
    if (origin == cpu) matches->setMemoryState(cpu);
    return ok;
  }
};

}  // namespace ssrlcv
