// ssrlcv_amd/host/SIFT_FeatureFactory.hpp -- FeatureFactory / SIFT_FeatureFactory with the reference's signatures
// (include/FeatureFactory.cuh:30-37,186-212, include/SIFT_FeatureFactory.cuh:53-72), bound to the HIP C ABI.
//
// generateFeatures keeps upstream's contract: pixels are forced onto the gpu for the call and restored to their origin
// state before returning; the result is a fresh Unity<Feature<SIFT_Descriptor>> in state gpu; zero key points is
// logger.err + exit(0) (src/SIFT_FeatureFactory.cu:118-121).  The whole sparse branch is one asynchronous C-ABI call
// (ssrlcv_hip_sift_extract) followed by ONE synchronisation to learn the feature count; upstream synchronises after
// each of its ~200 launches.  dense = true (never used by the pipeline, src/Pipeline.cu:25,44) is not provided.
#pragma once
#include <vector>
#include "Feature.hpp"
#include "Image.hpp"

namespace ssrlcv {

class FeatureFactory {
 protected:
  float orientationContribWidth;
  float descriptorContribWidth;

 public:
  FeatureFactory(float orientationContribWidth = 1.5f, float descriptorContribWidth = 6.0f)
      : orientationContribWidth(orientationContribWidth), descriptorContribWidth(descriptorContribWidth) {}
  void setOrientationContribWidth(float w) { orientationContribWidth = w; }
  void setDescriptorContribWidth(float w) { descriptorContribWidth = w; }
};

class SIFT_FeatureFactory : public FeatureFactory {
 public:
  SIFT_FeatureFactory(float orientationContribWidth = 1.5f, float descriptorContribWidth = 6.0f)
      : FeatureFactory(orientationContribWidth, descriptorContribWidth) {}

  ptr::value<Unity<Feature<SIFT_Descriptor>>> generateFeatures(ptr::value<Image> image, bool dense,
                                                               unsigned int maxOrientations,
                                                               float orientationThreshold = 0.8) {
    logger.info.printf("Generating SIFT features for image %d", image->id);
    if (dense) {
      logger.err << "ERROR: dense SIFT is not part of the MI355X hot path (the pipeline always passes dense = false)";
      std::exit(-1);
    }
    if (image->colorDepth != 1) {
      logger.err << "ERROR: only single-channel images are supported (convertToBW is outside the hot path)";
      std::exit(-1);
    }
    MemoryState origin = image->pixels->getMemoryState();
    if (origin != gpu) image->pixels->setMemoryState(gpu);

    ssrlcv_sift_params params;
    params.maxOrientations = maxOrientations;
    params.orientationThreshold = orientationThreshold;
    params.orientationContribWidth = this->orientationContribWidth;
    params.descriptorContribWidth = this->descriptorContribWidth;
    params.maxKeyPointsPerOctave = 0;
    ssrlcv_sift_plan* plan = nullptr;
    int rc = ssrlcv_sift_plan_create(image->size.x, image->size.y, &params, &plan);
    if (rc == SSRLCV_ERR_INVALID_ARG) {
      logger.err << "This image is too small to make a ScaleSpace of the specified depth";  // FeatureFactory.cu:343
      std::exit(-1);
    }
    HipSafeCall(rc);
    size_t wsBytes = ssrlcv_sift_plan_workspace_bytes(plan);
    uint32_t cap = ssrlcv_sift_plan_max_features(plan);
    ptr::device<unsigned char> workspace((long)wsBytes);
    ptr::device<Feature<SIFT_Descriptor>> all((long)cap);
    ptr::device<uint32_t> count_d(1);
    HipSafeCall(ssrlcv_hip_sift_extract(plan, image->pixels->device.get(), workspace.get(),
                                        reinterpret_cast<ssrlcv_sift_feature*>(all.get()), count_d.get(), nullptr));
    HipCheckError();
    uint32_t count = 0;
    HipSafeCall(ssrlcv_hip_memcpy(&count, count_d.get(), sizeof count, 1));
    ssrlcv_sift_plan_destroy(plan);
    if (origin != gpu) image->pixels->setMemoryState(origin);
    if (count == 0) {
      logger.err << "ERROR: something went wrong and there are 0 keypoints";
      std::exit(0);
    }
    if (count >= cap) {
      logger.err << "ERROR: key-point capacity exceeded";
      std::exit(-1);
    }
    logger.info.printf("total keypoints found = %d", count);
    ptr::device<Feature<SIFT_Descriptor>> exact((long)count);
    HipSafeCall(ssrlcv_hip_memcpy(exact.get(), all.get(), (size_t)count * sizeof(Feature<SIFT_Descriptor>), 2));
    return ptr::value<Unity<Feature<SIFT_Descriptor>>>(exact, (unsigned long)count, gpu);
  }
};

}  // namespace ssrlcv
