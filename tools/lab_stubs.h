// tools/lab_stubs.h -- the lab programs include pyramid.hip alone; build_dog's calls into keypoints.hip are never reached
// there and only need to link.
#pragma once
namespace svp {
void launch_polar_octave(const ssrlcv_sift_plan*, char*, int, hipStream_t) {}
int launch_chain_octave(const ssrlcv_sift_plan*, char*, int, hipStream_t) { return SSRLCV_ERR_UNSUPPORTED; }
}  // namespace svp
