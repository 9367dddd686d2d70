// ssrlcv_amd/host/Pipeline.hpp -- stage hand-off layer over the MI355X factories.
//
// The reference drives its six stages through plain structs (include/Pipeline.cuh:12-136) that a driver fills either
// from checkpoint files or from the previous stage's output (src/SFM.cu:131-230, test/Pipeline.cu).  The struct names,
// their fields and the fromCheckpoint / fromPreviousStage / do* signatures are that interface and are kept; how a
// stage is carried out is this build's own:
//   * everything a stage reads from disk goes through one reader (stage::Store) that knows the reference's file naming
//     (src/Pipeline.cu:3-10: "<dir>/<id>_<typeid>.uty", "<dir>/<id>_<typeid(Image)>.cpimg");
//   * the 2-view MatchSet (src/Pipeline.cu:198-224) is assembled ON THE DEVICE from the validated
//     DMatch list by ssrlcv_hip_matchset_from_matches, which also reduces the largest descriptor distance -- upstream
//     copies the list to the host twice (once for a max loop, once sliced to Match) and fills both arrays in a loop;
//   * the relative-pose arithmetic of the pose stage lives in two small functions that the tests can call.
// Out of scope at this edge: image decoding (images arrive as `.cpimg` + `<id>_h.uty` pixel
// checkpoints, or as ready Image objects), PoseEstimator::estimatePoseRANSAC (disabled upstream too), MeshFactory
// (its setPoints + savePoints pair is the ASCII PLY dump: writePLY).
#pragma once
#include <sys/stat.h>
#include <iomanip>
#include <sstream>
#include <string>
#include <typeinfo>
#include <vector>
#include "Image.hpp"
#include "MatchFactory.hpp"
#include "PointCloudFactory.hpp"
#include "PoseEstimator.hpp"
#include "SIFT_FeatureFactory.hpp"
#include "io_util.hpp"

#ifndef GEO_ORBIT
#define GEO_ORBIT 1  // the reference's Makefile default: earth-orbit epipolar constraint in the matcher
#endif

namespace ssrlcv {

typedef ptr::value<Unity<Feature<SIFT_Descriptor>>> SiftFeatures;

// PLY files of the cloud-producing stages land here ("out/" upstream); nothing is written when it does not exist
inline std::string& pipelineOutputDir() {
  static std::string where = "out/";
  return where;
}

namespace stage {

// A stage brackets its work with two marks of the same name in the reference's log; here the pair is one object.
class Marks {
  const char* name;
 public:
  explicit Marks(const char* n) : name(n) { logger.logState(name); }
  ~Marks() { logger.logState(name); }
  Marks(const Marks&) = delete;
  Marks& operator=(const Marks&) = delete;
};

inline bool onDisk(const std::string& file) {
  struct stat s;
  return ::stat(file.c_str(), &s) == 0;
}

// One checkpoint directory in the reference's naming scheme.
class Store {
  std::string root;
 public:
  explicit Store(const std::string& directory) : root(directory + "/") {}
  template <typename T> std::string unity(int id) const { return root + std::to_string(id) + "_" + typeid(T).name() + ".uty"; }
  std::string image(int id) const { return root + std::to_string(id) + "_" + typeid(Image).name() + ".cpimg"; }

  template <typename T> ptr::value<Unity<T>> load(int id) const { return ptr::value<Unity<T>>(unity<T>(id)); }
  template <typename T> ptr::value<Unity<T>> loadIfPresent(int id) const {
    const std::string file = unity<T>(id);
    return onDisk(file) ? ptr::value<Unity<T>>(file) : ptr::value<Unity<T>>(nullptr);
  }
  std::vector<ptr::value<Image>> cameras(int count) const {
    std::vector<ptr::value<Image>> v;
    v.reserve((size_t)count);
    for (int id = 0; id < count; ++id) v.emplace_back(image(id), id, true);
    return v;
  }
  std::vector<SiftFeatures> features(int count) const {
    std::vector<SiftFeatures> v;
    v.reserve((size_t)count);
    for (int id = 0; id < count; ++id) v.push_back(load<Feature<SIFT_Descriptor>>(id));
    return v;
  }
  MatchSet matchSet(int id = 0) const {
    MatchSet ms;
    ms.keyPoints = load<KeyPoint>(id);
    ms.matches = load<MultiMatch>(id);
    return ms;
  }
};

inline void dumpCloud(const char* name, const ptr::value<Unity<float3>>& cloud) {
  const std::string& dir = pipelineOutputDir();
  if (cloud != nullptr && onDisk(dir)) writePLY(name, cloud, dir);
}

inline void reportError(const std::string& label, float value) {
  std::ostringstream line;
  line << label << std::fixed << std::setprecision(12) << value;
  logger.info << line.str();
}

// Pose of camera `b` seen from camera `a`: position = (a - b) turned back through a's z, y, x rotations, angles = the
// axis rotations of Ra^T Rb (the starting point the reference hands to its LM refinement, src/Pipeline.cu:103-119).
inline Pose relativePose(const Image::Camera& a, const Image::Camera& b) {
  float3 t = a.cam_pos - b.cam_pos;
  const float3 axes[3] = {{0, 0, 1}, {0, 1, 0}, {1, 0, 0}};
  const float turn[3] = {-a.cam_rot.z, -a.cam_rot.y, -a.cam_rot.x};
  for (int k = 0; k < 3; ++k) t = rotatePointArbitrary(t, axes[k], turn[k]);
  float Ra[3][3], RaT[3][3], Rb[3][3], Rab[3][3];
  getRotationMatrix(a.cam_rot, Ra);
  transpose(Ra, RaT);
  getRotationMatrix(b.cam_rot, Rb);
  multiply(RaT, Rb, Rab);
  const float3 angles = getAxisRotations(Rab);
  Pose p;
  p.roll = angles.x;
  p.pitch = angles.y;
  p.yaw = angles.z;
  p.x = t.x;
  p.y = t.y;
  p.z = t.z;
  return p;
}

// The inverse step: put camera `b` where `pose` (relative to `a`, position in units of 1000) says it is (:127-136).
inline void applyRelativePose(const Image::Camera& a, const Pose& pose, Image::Camera& b) {
  const float3 step = {1000 * pose.x, 1000 * pose.y, 1000 * pose.z};
  b.cam_pos = a.cam_pos + rotatePoint(step, a.cam_rot);
  float Rrel[3][3], Ra[3][3], Rb[3][3];
  getRotationMatrix({pose.roll, pose.pitch, pose.yaw}, Rrel);
  getRotationMatrix(a.cam_rot, Ra);
  multiply(Ra, Rrel, Rb);
  b.cam_rot = getAxisRotations(Rb);
}

// M7 on the device.  `pairs` holds validated DMatches (any memory state); the result is left on the host, where the
// reference builds it and where its bundle generator expects it (src/PointCloudFactory.cu:840-846).
inline MatchSet pairwiseMatchSet(const ptr::value<Unity<DMatch>>& pairs, float* largestDistance) {
  const unsigned long n = pairs->size();
  const MemoryState before = pairs->getMemoryState();
  if (before == cpu) pairs->setMemoryState(gpu);
  else if (before == both && pairs->getFore() == cpu) pairs->transferMemoryTo(gpu);
  MatchSet ms;
  ms.keyPoints = ptr::value<Unity<KeyPoint>>(nullptr, 2 * n, gpu);
  ms.matches = ptr::value<Unity<MultiMatch>>(nullptr, n, gpu);
  ptr::device<float> peak(1);
  HipSafeCall(ssrlcv_hip_matchset_from_matches(SSRLCV_OUT_DMATCH, pairs->device.get(), (uint32_t)n,
                                               reinterpret_cast<ssrlcv_keypoint*>(ms.keyPoints->device.get()),
                                               reinterpret_cast<ssrlcv_multimatch*>(ms.matches->device.get()), peak.get(),
                                               nullptr));
  HipCheckError();
  HipSafeCall(ssrlcv_hip_memcpy(largestDistance, peak.get(), sizeof(float), 1));
  ms.keyPoints->setMemoryState(cpu);
  ms.matches->setMemoryState(cpu);
  if (before == cpu) pairs->setMemoryState(cpu);
  return ms;
}

}  // namespace stage

// =====================================================================================================================
// Stage 1: features (include/Pipeline.cuh:16-28)
struct FeatureGenerationInput {
  const std::string seedPath;
  const std::vector<std::string> imagePaths;
  const int numImages;
};

struct FeatureGenerationOutput {
  SiftFeatures seedFeatures;
  std::vector<SiftFeatures> allFeatures;
  std::vector<ptr::value<Image>> images;
};

// Images already in memory (pixels + camera).  Camera positions become relative to the first camera, whose ECEF
// position every image keeps as its offset (src/Pipeline.cu:36-39); features are left readable on the host.
inline void doFeatureGeneration(ptr::value<Image> seed, std::vector<ptr::value<Image>> images, FeatureGenerationOutput* out) {
  SIFT_FeatureFactory sift(1.5f, 6.0f);  // orientation / descriptor contribution widths of the reference pipeline
  const unsigned maxOrientations = 2;
  const float peakRatio = 0.8f;
  {
    stage::Marks marks("SEED");
    if (seed != nullptr) out->seedFeatures = sift.generateFeatures(seed, false, maxOrientations, peakRatio);
  }
  stage::Marks marks("FEATURES");
  if (images.empty()) return;
  const float3 origin = images.front()->camera.cam_pos;
  out->images.reserve(images.size());
  out->allFeatures.reserve(images.size());
  for (ptr::value<Image>& view : images) {
    view->camera.ecef_offset = origin;
    view->camera.cam_pos = view->camera.cam_pos - origin;
    SiftFeatures found = sift.generateFeatures(view, false, maxOrientations, peakRatio);
    found->transferMemoryTo(cpu);
    out->allFeatures.push_back(found);
    out->images.push_back(view);
  }
}

// Path flavour.  This build has no image decoders: a path must name a camera checkpoint `<dir>/<id>_<Image>.cpimg`
// with the pixels beside it as `<dir>/<id>_h.uty`.
inline void doFeatureGeneration(FeatureGenerationInput* in, FeatureGenerationOutput* out) {
  auto open = [](const std::string& cameraFile, int id) {
    const std::string dir = cameraFile.substr(0, cameraFile.find_last_of('/') + 1);
    const std::string pixelFile = dir + std::to_string(id) + "_h.uty";
    if (!stage::onDisk(pixelFile)) {
      logger.err << "no pixel checkpoint " + pixelFile + " (this build reads .cpimg + _h.uty pairs, it decodes no image formats)";
      std::exit(-1);
    }
    ptr::value<Image> view(cameraFile, id, true);
    view->pixels = ptr::value<Unity<unsigned char>>(pixelFile);
    return view;
  };
  std::vector<ptr::value<Image>> views;
  for (int id = 0; id < in->numImages; ++id) views.push_back(open(in->imagePaths[(size_t)id], id));
  ptr::value<Image> seed = in->seedPath.empty() ? ptr::value<Image>(nullptr) : open(in->seedPath, -1);
  doFeatureGeneration(seed, views, out);
}

// =====================================================================================================================
// Stage 2: pose (include/Pipeline.cuh:34-50)
struct PoseEstimationInput {
  SiftFeatures seedFeatures;
  std::vector<SiftFeatures> allFeatures;
  std::vector<ptr::value<Image>> images;

  void fromCheckpoint(std::string featureGenDir, int numImages) {
    const stage::Store store(featureGenDir);
    seedFeatures = store.loadIfPresent<Feature<SIFT_Descriptor>>(-1);
    images = store.cameras(numImages);
    allFeatures = store.features(numImages);
  }
  void fromPreviousStage(FeatureGenerationOutput* featureGenOutput) {
    seedFeatures = featureGenOutput->seedFeatures;
    allFeatures = featureGenOutput->allFeatures;
    images = featureGenOutput->images;
  }
};

struct PoseEstimationOutput {
  ptr::value<Unity<float>> seedDistances = nullptr;
};

// Two views only, like upstream: loose double-constrained matches (epsilon 100 px, delta 3 km, absolute threshold 10^2)
// feed the Levenberg-Marquardt refinement, which starts at the cameras' own relative pose; the second camera is then
// moved to the refined pose.
inline void doPoseEstimation(PoseEstimationInput* in, PoseEstimationOutput* out) {
  stage::Marks marks("POSE");
  ptr::value<Image> first = in->images.at(0), second = in->images.at(1);
  MatchFactory<SIFT_Descriptor> matcher(0.6f, 10.0f * 10.0f);
  const bool seeded = in->seedFeatures != nullptr;
  if (seeded) {
    matcher.setSeedFeatures(in->seedFeatures);
    out->seedDistances = matcher.getSeedDistances(in->allFeatures[0]);
  } else {
    out->seedDistances = nullptr;
  }
  ptr::value<Unity<Match>> tiePoints = matcher.generateMatchesDoubleConstrained(first, in->allFeatures[0], second, in->allFeatures[1],
                                                                               100, 3, out->seedDistances);
  tiePoints->transferMemoryTo(cpu);
  logger.info.printf("pose stage: %lu tie points", tiePoints->size());
  PoseEstimator refiner(first, second, tiePoints);
  Pose pose = stage::relativePose(first->camera, second->camera);
  logger.info.printf("pose stage: start angles %f %f %f", pose.roll, pose.pitch, pose.yaw);
  refiner.LM_optimize(&pose);
  stage::applyRelativePose(first->camera, pose, second->camera);
  const Image::Camera& moved = second->camera;
  logger.info.printf("pose stage: camera 1 now at (%f %f %f), rotation (%f %f %f)", moved.cam_pos.x, moved.cam_pos.y,
                     moved.cam_pos.z, moved.cam_rot.x, moved.cam_rot.y, moved.cam_rot.z);
}

// =====================================================================================================================
// Stage 3: matching (include/Pipeline.cuh:56-74)
struct FeatureMatchingInput {
  SiftFeatures seedFeatures;
  std::vector<SiftFeatures> allFeatures;
  std::vector<ptr::value<Image>> images;
  ptr::value<Unity<float>> seedDistances;
  float epsilon;  // pixels either side of the projected epipolar segment
  float delta;    // kilometres added above / below the earth shells that bound the segment

  void fromCheckpoint(std::string featureGenDirectory, std::string poseDirectory, int numImages, float epsilon, float delta) {
    const stage::Store store(featureGenDirectory);
    seedFeatures = store.loadIfPresent<Feature<SIFT_Descriptor>>(-1);
    images = store.cameras(numImages);
    allFeatures = store.features(numImages);
    // seed distances are a product of the (2-view only) pose stage
    if (numImages == 2) seedDistances = stage::Store(poseDirectory).loadIfPresent<float>(0);
    this->epsilon = epsilon;
    this->delta = delta;
  }
  void fromPreviousStage(PoseEstimationInput* poseInput, PoseEstimationOutput* poseOutput, float epsilon, float delta) {
    seedFeatures = poseInput->seedFeatures;
    allFeatures = poseInput->allFeatures;
    images = poseInput->images;
    seedDistances = poseOutput->seedDistances;
    this->epsilon = epsilon;
    this->delta = delta;
  }
};

struct FeatureMatchingOutput {
  MatchSet matchSet;
};

// Ratio threshold 0.6 against the seed image, absolute threshold 200^2 (src/Pipeline.cu:175).  Two views: one matcher
// call, MatchSet assembled on the device.  More: every pair i < j + the consistency merge (generateMatchesExhaustive).
inline void doFeatureMatching(FeatureMatchingInput* in, FeatureMatchingOutput* out) {
  stage::Marks marks("MATCHING");
  MatchFactory<SIFT_Descriptor> matcher(0.6f, 200.0f * 200.0f);
  if (in->seedFeatures != nullptr) {
    matcher.setSeedFeatures(in->seedFeatures);
    if (in->seedDistances == nullptr) in->seedDistances = matcher.getSeedDistances(in->allFeatures[0]);
  }
  if (in->images.size() != 2) {
    out->matchSet = matcher.generateMatchesExhaustive(in->images, in->allFeatures, in->epsilon, in->delta);
    out->matchSet.keyPoints->setMemoryState(cpu);
    out->matchSet.matches->setMemoryState(cpu);
    return;
  }
  ptr::value<Unity<DMatch>> pairs =
#if GEO_ORBIT == 1
      matcher.generateDistanceMatchesDoubleConstrained(in->images[0], in->allFeatures[0], in->images[1], in->allFeatures[1],
                                                       in->epsilon, in->delta, in->seedDistances);
#else
      matcher.generateDistanceMatches(in->images[0], in->allFeatures[0], in->images[1], in->allFeatures[1], in->seedDistances);
#endif
  float worst = 0.0f;
  out->matchSet = stage::pairwiseMatchSet(pairs, &worst);
  logger.info.printf("2-view matching: %lu matches, largest descriptor distance %f", out->matchSet.matches->size(), worst);
}

// =====================================================================================================================
// Stage 4: triangulation (include/Pipeline.cuh:80-92)
struct TriangulationInput {
  MatchSet matchSet;
  std::vector<ptr::value<Image>> images;

  void fromCheckpoint(std::string featureGenDir, std::string featureMatchDir, int numImages) {
    images = stage::Store(featureGenDir).cameras(numImages);
    matchSet = stage::Store(featureMatchDir).matchSet();
  }
  void fromPreviousStage(FeatureMatchingInput* featureMatchingInput, FeatureMatchingOutput* featureMatchingOutput) {
    images = featureMatchingInput->images;
    matchSet = featureMatchingOutput->matchSet;
  }
};

struct TriangulationOutput {
  ptr::value<Unity<float3>> points;
};

namespace stage {
// bundles -> cloud with the triangulator the view count selects; `error` is the linear (2-view) or angular (N-view) sum
inline ptr::value<Unity<float3>> triangulate(PointCloudFactory& factory, MatchSet* matchSet,
                                             const std::vector<ptr::value<Image>>& views, float* error) {
  BundleSet rays = factory.generateBundles(matchSet, views);
  if (views.size() == 2) return factory.twoViewTriangulate(rays, error);
  return factory.nViewTriangulate(rays, error);
}
}  // namespace stage

inline void doTriangulation(TriangulationInput* in, TriangulationOutput* out) {
  stage::Marks marks("TRIANGULATE");
  PointCloudFactory factory;
  float error = 0.0f;
  out->points = stage::triangulate(factory, &in->matchSet, in->images, &error);
  stage::reportError("triangulation error before filtering: ", error);
  stage::dumpCloud("ssrlcv-initial", out->points);
}

// =====================================================================================================================
// Stage 5: filtering (include/Pipeline.cuh:98-110)
struct FilteringInput {
  MatchSet matchSet;
  std::vector<ptr::value<Image>> images;

  void fromCheckpoint(std::string featureGenDir, std::string featureMatchDir, int numImages) {
    images = stage::Store(featureGenDir).cameras(numImages);
    matchSet = stage::Store(featureMatchDir).matchSet();
  }
  void fromPreviousStage(TriangulationInput* triangulationInput) {
    images = triangulationInput->images;
    matchSet = triangulationInput->matchSet;
  }
};

struct FilteringOutput {
  ptr::value<Unity<float3>> points;
};

// Two views: drop matches whose rays miss each other by more than 100 km, then everything beyond 3 sigma of a 10 %
// sample; N views: the statistical filter alone (src/Pipeline.cu:297-343).  The filters rewrite in->matchSet; the
// cloud is triangulated again from what is left.
inline void doFiltering(FilteringInput* in, FilteringOutput* out) {
  stage::Marks marks("FILTER");
  PointCloudFactory factory;
  const float sigmas = 3.0f, sampleFraction = 0.1f;
  if (in->images.size() == 2) factory.linearCutoffFilter(&in->matchSet, in->images, 100.0);
  factory.deterministicStatisticalFilter(&in->matchSet, in->images, sigmas, sampleFraction);
  float error = 0.0f;
  out->points = stage::triangulate(factory, &in->matchSet, in->images, &error);
  stage::reportError("triangulation error after the " + std::to_string((int)sigmas) + "-sigma filter: ", error);
  stage::dumpCloud("ssrlcv-filtered", out->points);
}

// =====================================================================================================================
// Stage 6: bundle adjustment (include/Pipeline.cuh:116-134)
struct BundleAdjustInput {
  MatchSet matchSet;
  std::vector<ptr::value<Image>> images;

  void fromCheckpoint(std::string featureGenDir, std::string filteringDir, int numImages) {
    images = stage::Store(featureGenDir).cameras(numImages);
    matchSet = stage::Store(filteringDir).matchSet();
  }
  void fromPreviousStage(FilteringInput* filteringInput) {
    images = filteringInput->images;
    matchSet = filteringInput->matchSet;
  }
};

struct BundleAdjustOutput {
  ptr::value<Unity<float3>> points;
};

// Ten iterations requested, two views only (upstream has no N-view bundle adjustment: the call returns without a cloud).
inline void doBundleAdjust(BundleAdjustInput* in, BundleAdjustOutput* out) {
  if (in->images.size() != 2) return;
  stage::Marks marks("BA");
  PointCloudFactory factory;
  const unsigned iterations = 10;
  out->points = factory.BundleAdjustTwoView(&in->matchSet, in->images, iterations, "");
  stage::dumpCloud("ssrlcv-BA-final", out->points);
}

}  // namespace ssrlcv
