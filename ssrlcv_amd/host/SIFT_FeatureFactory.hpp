// ssrlcv_amd/host/SIFT_FeatureFactory.hpp -- FeatureFactory / SIFT_FeatureFactory with the reference's signatures
// (include/FeatureFactory.cuh:30-37,186-212, include/SIFT_FeatureFactory.cuh:53-72), bound to the HIP C ABI.
//
// generateFeatures keeps upstream's contract: pixels are forced onto the gpu for the call and restored to their origin
// state before returning; the result is a fresh Unity<Feature<SIFT_Descriptor>> in state gpu; zero key points is
// logger.err + exit(0) (src/SIFT_FeatureFactory.cu:118-121).  The whole sparse branch is one asynchronous C-ABI call
// (ssrlcv_hip_sift_extract) followed by ONE synchronisation to learn the feature count; upstream synchronises after
// each of its ~200 launches.  dense = true (never used by the pipeline, src/Pipeline.cu:25,44) is not provided.
// Colour input goes through convertToBW like upstream (src/SIFT_FeatureFactory.cu:26-29).
#pragma once
#include <map>
#include <memory>
#include <tuple>
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
  // One plan + workspace + staging feature buffer per (image size, parameters), kept for the factory's lifetime (and
  // shared by its copies): doFeatureGeneration runs every image of a set through one factory, and creating the plan,
  // hipMalloc-ing ~4.5 GB of workspace and 1.6 GB of feature staging and freeing them again PER IMAGE cost more than the
  // extraction itself.  The reference allocates per call too, but through cudaMalloc of exact-size lists.
  struct Slot {
    ssrlcv_sift_plan* plan = nullptr;
    ptr::device<unsigned char> workspace;
    ptr::device<Feature<SIFT_Descriptor>> staging;
    ptr::device<uint32_t> count;
    uint32_t cap = 0;
    uint32_t perOctave = 0;  // 0 = the plan's default density bound
    ~Slot() {
      if (plan) ssrlcv_sift_plan_destroy(plan);
    }
  };
  typedef std::tuple<unsigned, unsigned, unsigned, float, float, float> Key;
  std::shared_ptr<std::map<Key, std::shared_ptr<Slot>>> pool;

  void build(Slot& slot, uint2 size, const ssrlcv_sift_params& params) {
    if (slot.plan) ssrlcv_sift_plan_destroy(slot.plan);
    slot.plan = nullptr;
    int rc = ssrlcv_sift_plan_create(size.x, size.y, &params, &slot.plan);
    if (rc == SSRLCV_ERR_INVALID_ARG) {
      logger.err << "This image is too small to make a ScaleSpace of the specified depth";  // FeatureFactory.cu:343
      std::exit(-1);
    }
    if (rc == SSRLCV_ERR_UNSUPPORTED) {
      logger.err << "ERROR: image size / contribution widths outside what the MI355X SIFT plan supports "
                    "(input below 64 px on a side -- upstream's 65-tap mirror reads outside its smallest octave there -- or "
                    "octave 0 above 16384 px, descriptor width > 30, orientation width > 5)";
      std::exit(-1);
    }
    HipSafeCall(rc);
    slot.cap = ssrlcv_sift_plan_max_features(slot.plan);
    slot.workspace = ptr::device<unsigned char>((long)ssrlcv_sift_plan_workspace_bytes(slot.plan));
    slot.staging = ptr::device<Feature<SIFT_Descriptor>>((long)slot.cap);
    if (!slot.count.get()) slot.count = ptr::device<uint32_t>(1);
  }

 public:
  SIFT_FeatureFactory(float orientationContribWidth = 1.5f, float descriptorContribWidth = 6.0f)
      : FeatureFactory(orientationContribWidth, descriptorContribWidth),
        pool(std::make_shared<std::map<Key, std::shared_ptr<Slot>>>()) {}

  // drops the cached plans / workspaces (they are also released with the last copy of the factory)
  void releaseWorkspaces() { pool->clear(); }

  ptr::value<Unity<Feature<SIFT_Descriptor>>> generateFeatures(ptr::value<Image> image, bool dense,
                                                               unsigned int maxOrientations,
                                                               float orientationThreshold = 0.8) {
    logger.info.printf("Generating SIFT features for image %d", image->id);
    if (dense) {
      logger.err << "ERROR: dense SIFT is not part of the MI355X hot path (the pipeline always passes dense = false)";
      std::exit(-1);
    }
    MemoryState origin = image->pixels->getMemoryState();
    // Upstream moves the pixels to the device with a hard setMemoryState(gpu) and back with setMemoryState(origin)
    // (src/SIFT_FeatureFactory.cu:22-24,166): the second is a device-to-host copy of bytes the host had a moment ago.  The
    // kernels only READ single-channel pixels, so for cpu-state grey pixels the host copy is kept (transferMemoryTo: state
    // both) and the way back is clear(gpu): the same state and the same bytes at return, without the copy (0.55 ms of a
    // 4096^2 image's 6.1).  Colour pixels are converted on the device and take upstream's round trip.
    const bool keepHost = origin == cpu && image->colorDepth == 1;
    if (keepHost) image->pixels->transferMemoryTo(gpu);
    else if (origin != gpu) image->pixels->setMemoryState(gpu);
    // convert image to BW (src/SIFT_FeatureFactory.cu:26-29)
    if (image->colorDepth != 1) {
      convertToBW(image->pixels, image->colorDepth);
      image->colorDepth = 1;
    }

    ssrlcv_sift_params params;
    params.maxOrientations = maxOrientations;
    params.orientationThreshold = orientationThreshold;
    params.orientationContribWidth = this->orientationContribWidth;
    params.descriptorContribWidth = this->descriptorContribWidth;
    params.maxKeyPointsPerOctave = 0;
    const Key key(image->size.x, image->size.y, maxOrientations, orientationThreshold, this->orientationContribWidth,
                  this->descriptorContribWidth);
    std::shared_ptr<Slot>& slotRef = (*pool)[key];
    if (!slotRef) slotRef = std::make_shared<Slot>();
    Slot& slot = *slotRef;
    params.maxKeyPointsPerOctave = slot.perOctave;
    if (!slot.plan) build(slot, image->size, params);
    uint32_t count = 0;
    for (;;) {
      HipSafeCall(ssrlcv_hip_sift_extract(slot.plan, image->pixels->device.get(), slot.workspace.get(),
                                          reinterpret_cast<ssrlcv_sift_feature*>(slot.staging.get()), slot.count.get(),
                                          nullptr));
      HipCheckError();
      // The reference's lists are unbounded; the plan's are sized up front.  A list that outgrew its capacity was
      // truncated: grow and run again, so that the caller always gets the reference's result.
      uint32_t overflowMask = 0;
      int rc = ssrlcv_sift_plan_overflow(slot.plan, slot.workspace.get(), &overflowMask, nullptr);
      if (rc != SSRLCV_ERR_CAPACITY) {
        HipSafeCall(rc);
        break;
      }
      const unsigned long px = 4ul * image->size.x * image->size.y;
      slot.perOctave = slot.perOctave ? slot.perOctave * 2 : (uint32_t)(px / 8);  // default bound is px / 16
      logger.warn.printf("key-point capacity exceeded (octave mask %u): re-running with %u key points per octave",
                         overflowMask, slot.perOctave);
      params.maxKeyPointsPerOctave = slot.perOctave;
      build(slot, image->size, params);
    }
    HipSafeCall(ssrlcv_hip_memcpy(&count, slot.count.get(), sizeof count, 1));
    if (keepHost) image->pixels->clear(gpu);
    else if (origin != gpu) image->pixels->setMemoryState(origin);
    if (count == 0) {
      logger.err << "ERROR: something went wrong and there are 0 keypoints";
      std::exit(0);
    }
    logger.info.printf("total keypoints found = %d", count);
    ptr::device<Feature<SIFT_Descriptor>> exact((long)count);
    HipSafeCall(ssrlcv_hip_memcpy(exact.get(), slot.staging.get(), (size_t)count * sizeof(Feature<SIFT_Descriptor>), 2));
    return ptr::value<Unity<Feature<SIFT_Descriptor>>>(exact, (unsigned long)count, gpu);
  }
};

}  // namespace ssrlcv
