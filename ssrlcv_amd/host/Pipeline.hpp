// ssrlcv_amd/host/Pipeline.hpp -- the stage glue of the reference (include/Pipeline.cuh:12-136, src/Pipeline.cu:16-384)
// over the MI355X factories: the same stage structs, `fromCheckpoint` / `fromPreviousStage` hand-offs and `do*` entry
// points, so a driver written against the reference (src/SFM.cu:131-230, test/Pipeline.cu) compiles unchanged.
//
// Differences, all at the edges of the hot path (SURVEY.md section 8, rows marked out of scope):
//  - image decoding is not part of this build: doFeatureGeneration(FeatureGenerationInput*, ...) reads `.cpimg` camera
//    checkpoints plus `<id>_h.uty` pixel checkpoints when an image path names a directory entry of that form, and
//    otherwise fails like the reference does on an unreadable image; the overload taking ready `Image`s is what the
//    tests and the Python driver use;
//  - doPoseEstimation runs the reference's flow (seed distances, double-constrained matches, LM refinement from the
//    cameras' relative pose); PoseEstimator::estimatePoseRANSAC, commented out upstream, is not built;
//  - MeshFactory::setPoints + savePoints("name") is the ASCII PLY dump, done with writePLY.
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
#define GEO_ORBIT 1  // Makefile default of the reference
#endif

namespace ssrlcv {

typedef ptr::value<Unity<Feature<SIFT_Descriptor>>> SiftFeatures;

namespace detail {
template <typename T>
inline std::string checkpointPath(const std::string& directory, int id) {  // src/Pipeline.cu:3-6
  return directory + "/" + std::to_string(id) + "_" + typeid(T).name() + ".uty";
}
inline std::string imageCheckpointPath(const std::string& directory, int id) {  // src/Pipeline.cu:8-10
  return directory + "/" + std::to_string(id) + "_" + typeid(Image).name() + ".cpimg";
}
inline bool exists(const std::string& path) {
  struct stat buf;
  return stat(path.c_str(), &buf) == 0;
}
inline std::vector<ptr::value<Image>> imagesFromCheckpoint(const std::string& dir, int numImages) {
  std::vector<ptr::value<Image>> images;
  for (int i = 0; i < numImages; i++) images.push_back(ptr::value<Image>(imageCheckpointPath(dir, i), i, true));
  return images;
}
inline void savePoints(const char* name, ptr::value<Unity<float3>> points, const std::string& dir) {
  if (points != nullptr && exists(dir)) writePLY(name, points, dir);
}
}  // namespace detail

// where the do* stages drop their PLY files (the reference hard-wires "out/"; skipped when the directory is absent)
inline std::string& pipelineOutputDir() {
  static std::string dir = "out/";
  return dir;
}

// ---------------------------------------------------------------------------------------------------------------------
// FEATURE GENERATION (include/Pipeline.cuh:16-28, src/Pipeline.cu:16-51)
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

// the reference's loop body for images that are already in memory (pixels in image->pixels, camera filled in)
inline void doFeatureGeneration(ptr::value<Image> seed, std::vector<ptr::value<Image>> images, FeatureGenerationOutput* out) {
  SIFT_FeatureFactory featureFactory = SIFT_FeatureFactory(1.5f, 6.0f);
  logger.logState("SEED");
  if (seed != nullptr) out->seedFeatures = featureFactory.generateFeatures(seed, false, 2, 0.8);
  logger.logState("SEED");
  logger.logState("FEATURES");
  float3 offset = {0.0f, 0.0f, 0.0f};
  for (size_t i = 0; i < images.size(); i++) {
    ptr::value<Image> image = images[i];
    if (i == 0) offset = image->camera.cam_pos;
    image->camera.ecef_offset = offset;
    image->camera.cam_pos = image->camera.cam_pos - offset;
    SiftFeatures features = featureFactory.generateFeatures(image, false, 2, 0.8);
    features->transferMemoryTo(cpu);
    out->images.push_back(image);
    out->allFeatures.push_back(features);
  }
  logger.logState("FEATURES");
}

// path flavour: every path must be a `<dir>/<id>_<typeid(Image)>.cpimg` checkpoint with `<dir>/<id>_h.uty` pixels beside it
inline void doFeatureGeneration(FeatureGenerationInput* in, FeatureGenerationOutput* out) {
  auto load = [](const std::string& path, int id) {
    ptr::value<Image> image(path, id, true);
    std::string pix = path.substr(0, path.find_last_of('/') + 1) + std::to_string(id) + "_h.uty";
    if (!detail::exists(pix)) {
      logger.err << "image decoding is not part of this build and no pixel checkpoint " + pix + " exists";
      exit(-1);
    }
    image->pixels = ptr::value<Unity<unsigned char>>(pix);
    return image;
  };
  ptr::value<Image> seed = nullptr;
  if (in->seedPath.size() > 0) seed = load(in->seedPath, -1);
  std::vector<ptr::value<Image>> images;
  for (int i = 0; i < in->numImages; i++) images.push_back(load(in->imagePaths[i], i));
  doFeatureGeneration(seed, images, out);
}

// ---------------------------------------------------------------------------------------------------------------------
// POSE ESTIMATION (include/Pipeline.cuh:34-50, src/Pipeline.cu:57-140)
struct PoseEstimationInput {
  SiftFeatures seedFeatures;
  std::vector<SiftFeatures> allFeatures;
  std::vector<ptr::value<Image>> images;

  void fromCheckpoint(std::string featureGenDir, int numImages) {
    std::string seedCpPath = detail::checkpointPath<Feature<SIFT_Descriptor>>(featureGenDir, -1);
    if (detail::exists(seedCpPath)) seedFeatures = SiftFeatures(seedCpPath);
    images = detail::imagesFromCheckpoint(featureGenDir, numImages);
    for (int i = 0; i < numImages; i++)
      allFeatures.push_back(SiftFeatures(detail::checkpointPath<Feature<SIFT_Descriptor>>(featureGenDir, i)));
  }
  void fromPreviousStage(FeatureGenerationOutput* featureGenOutput) {
    this->seedFeatures = featureGenOutput->seedFeatures;
    this->allFeatures = featureGenOutput->allFeatures;
    this->images = featureGenOutput->images;
  }
};

struct PoseEstimationOutput {
  ptr::value<Unity<float>> seedDistances = nullptr;
};

inline void doPoseEstimation(PoseEstimationInput* in, PoseEstimationOutput* out) {
  logger.info << "Starting pose estimation...";
  logger.logState("POSE");
  MatchFactory<SIFT_Descriptor> matchFactory = MatchFactory<SIFT_Descriptor>(0.6f, 10.0f * 10.0f);
  if (in->seedFeatures != nullptr) matchFactory.setSeedFeatures(in->seedFeatures);
  out->seedDistances = (in->seedFeatures != nullptr) ? matchFactory.getSeedDistances(in->allFeatures[0]) : nullptr;
  logger.logState("matching images");
  ptr::value<Unity<Match>> matches = matchFactory.generateMatchesDoubleConstrained(
      in->images[0], in->allFeatures[0], in->images[1], in->allFeatures[1], 100, 3, out->seedDistances);
  logger.logState("done matching images");
  matches->transferMemoryTo(cpu);
  PoseEstimator estim(in->images.at(0), in->images.at(1), matches);
  // starting pose = the cameras' relative pose (the RANSAC estimate is commented out upstream, src/Pipeline.cu:101)
  Pose pose;
  float3 pos = in->images.at(0)->camera.cam_pos - in->images.at(1)->camera.cam_pos;
  pos = rotatePointArbitrary(pos, {0, 0, 1}, -in->images.at(0)->camera.cam_rot.z);
  pos = rotatePointArbitrary(pos, {0, 1, 0}, -in->images.at(0)->camera.cam_rot.y);
  pos = rotatePointArbitrary(pos, {1, 0, 0}, -in->images.at(0)->camera.cam_rot.x);
  pose.x = pos.x;
  pose.y = pos.y;
  pose.z = pos.z;
  float C0[3][3], C0t[3][3], C1[3][3], relative[3][3];
  getRotationMatrix(in->images.at(0)->camera.cam_rot, C0);
  transpose(C0, C0t);
  getRotationMatrix(in->images.at(1)->camera.cam_rot, C1);
  multiply(C0t, C1, relative);
  float3 rot = getAxisRotations(relative);
  pose.roll = rot.x;
  pose.pitch = rot.y;
  pose.yaw = rot.z;
  logger.info.printf("Original pose: %f %f %f", pose.roll, pose.pitch, pose.yaw);
  estim.LM_optimize(&pose);
  // write the refined pose back into the second camera (src/Pipeline.cu:127-136)
  float R1[3][3], R2[3][3], R[3][3];
  in->images.at(1)->camera.cam_pos =
      in->images.at(0)->camera.cam_pos + rotatePoint({1000 * pose.x, 1000 * pose.y, 1000 * pose.z}, in->images.at(0)->camera.cam_rot);
  getRotationMatrix({pose.roll, pose.pitch, pose.yaw}, R1);
  getRotationMatrix(in->images.at(0)->camera.cam_rot, R2);
  multiply(R2, R1, R);
  in->images.at(1)->camera.cam_rot = getAxisRotations(R);
  logger.info.printf("Rotation: %f %f %f", in->images.at(1)->camera.cam_rot.x, in->images.at(1)->camera.cam_rot.y, in->images.at(1)->camera.cam_rot.z);
  logger.info.printf("Position: %f %f %f", in->images.at(1)->camera.cam_pos.x, in->images.at(1)->camera.cam_pos.y, in->images.at(1)->camera.cam_pos.z);
  logger.logState("POSE");
}

// ---------------------------------------------------------------------------------------------------------------------
// FEATURE MATCHING (include/Pipeline.cuh:56-74, src/Pipeline.cu:146-241)
struct FeatureMatchingInput {
  SiftFeatures seedFeatures;
  std::vector<SiftFeatures> allFeatures;
  std::vector<ptr::value<Image>> images;
  ptr::value<Unity<float>> seedDistances;
  float epsilon;  // pixel buffer around 2D epipolar line
  float delta;    // kilometer buffer above and below line segment in 3D space

  void fromCheckpoint(std::string featureGenDirectory, std::string poseDirectory, int numImages, float epsilon, float delta) {
    std::string seedCpPath = detail::checkpointPath<Feature<SIFT_Descriptor>>(featureGenDirectory, -1);
    if (detail::exists(seedCpPath)) seedFeatures = SiftFeatures(seedCpPath);
    images = detail::imagesFromCheckpoint(featureGenDirectory, numImages);
    for (int i = 0; i < numImages; i++)
      allFeatures.push_back(SiftFeatures(detail::checkpointPath<Feature<SIFT_Descriptor>>(featureGenDirectory, i)));
    this->epsilon = epsilon;
    this->delta = delta;
    if (numImages == 2) {  // pose estimation only exists for 2 views
      std::string seedDistPath = detail::checkpointPath<float>(poseDirectory, 0);
      if (detail::exists(seedDistPath)) seedDistances = ptr::value<Unity<float>>(seedDistPath);
    }
  }
  void fromPreviousStage(PoseEstimationInput* poseInput, PoseEstimationOutput* poseOutput, float epsilon, float delta) {
    this->seedFeatures = poseInput->seedFeatures;
    this->allFeatures = poseInput->allFeatures;
    this->images = poseInput->images;
    this->seedDistances = poseOutput->seedDistances;
    this->epsilon = epsilon;
    this->delta = delta;
  }
};

struct FeatureMatchingOutput {
  MatchSet matchSet;
};

inline void doFeatureMatching(FeatureMatchingInput* in, FeatureMatchingOutput* out) {
  logger.info << "Starting matching...";
  MatchFactory<SIFT_Descriptor> matchFactory = MatchFactory<SIFT_Descriptor>(0.6f, 200.0f * 200.0f);
  logger.logState("MATCHING");
  if (in->seedFeatures != nullptr) matchFactory.setSeedFeatures(in->seedFeatures);
  if (in->seedDistances == nullptr)
    in->seedDistances = (in->seedFeatures != nullptr) ? matchFactory.getSeedDistances(in->allFeatures[0]) : nullptr;
  if (in->images.size() == 2) {
#if GEO_ORBIT == 1
    ptr::value<Unity<DMatch>> distanceMatches = matchFactory.generateDistanceMatchesDoubleConstrained(
        in->images[0], in->allFeatures[0], in->images[1], in->allFeatures[1], in->epsilon, in->delta, in->seedDistances);
#else
    ptr::value<Unity<DMatch>> distanceMatches = matchFactory.generateDistanceMatches(
        in->images[0], in->allFeatures[0], in->images[1], in->allFeatures[1], in->seedDistances);
#endif
    distanceMatches->transferMemoryTo(cpu);
    float maxDist = 0.0f;
    DMatch* dhost = distanceMatches->host.get();
    for (unsigned long i = 0; i < distanceMatches->size(); ++i)
      if (maxDist < dhost[i].distance) maxDist = dhost[i].distance;
    logger.info.printf("max euclidean distance between features = %f", maxDist);
    if (distanceMatches->getMemoryState() != gpu) distanceMatches->setMemoryState(gpu);
    ptr::value<Unity<Match>> matches = matchFactory.getRawMatches(distanceMatches);
    // the 2-view MatchSet is the match list laid out pairwise (src/Pipeline.cu:204-223)
    out->matchSet.keyPoints = ptr::value<Unity<KeyPoint>>(nullptr, matches->size() * 2, cpu);
    out->matchSet.matches = ptr::value<Unity<MultiMatch>>(nullptr, matches->size(), cpu);
    matches->setMemoryState(cpu);
    KeyPoint* okhost = out->matchSet.keyPoints->host.get();
    Match* mhost = matches->host.get();
    MultiMatch* omhost = out->matchSet.matches->host.get();
    for (unsigned long i = 0; i < out->matchSet.matches->size(); i++) {
      okhost[i * 2] = mhost[i].keyPoints[0];
      okhost[i * 2 + 1] = mhost[i].keyPoints[1];
      omhost[i] = {2, (int)(i * 2)};
    }
    logger.info << "Total Matches: " + std::to_string(matches->size());
  } else {
    out->matchSet = matchFactory.generateMatchesExhaustive(in->images, in->allFeatures, in->epsilon, in->delta);
    out->matchSet.matches->setMemoryState(cpu);
    out->matchSet.keyPoints->setMemoryState(cpu);
  }
  logger.logState("MATCHING");
}

// ---------------------------------------------------------------------------------------------------------------------
// TRIANGULATION (include/Pipeline.cuh:80-92, src/Pipeline.cu:247-278)
struct TriangulationInput {
  MatchSet matchSet;
  std::vector<ptr::value<Image>> images;

  void fromCheckpoint(std::string featureGenDir, std::string featureMatchDir, int numImages) {
    images = detail::imagesFromCheckpoint(featureGenDir, numImages);
    matchSet.keyPoints = ptr::value<Unity<KeyPoint>>(detail::checkpointPath<KeyPoint>(featureMatchDir, 0));
    matchSet.matches = ptr::value<Unity<MultiMatch>>(detail::checkpointPath<MultiMatch>(featureMatchDir, 0));
  }
  void fromPreviousStage(FeatureMatchingInput* featureMatchingInput, FeatureMatchingOutput* featureMatchingOutput) {
    this->images = featureMatchingInput->images;
    this->matchSet = featureMatchingOutput->matchSet;
  }
};

struct TriangulationOutput {
  ptr::value<Unity<float3>> points;
};

inline void doTriangulation(TriangulationInput* in, TriangulationOutput* out) {
  PointCloudFactory pointCloudFactory = PointCloudFactory();
  logger.logState("TRIANGULATE");
  float error;  // linear for 2-view, angular for N-view
  BundleSet bundleSet = pointCloudFactory.generateBundles(&in->matchSet, in->images);
  out->points = (in->images.size() == 2) ? pointCloudFactory.twoViewTriangulate(bundleSet, &error)
                                         : pointCloudFactory.nViewTriangulate(bundleSet, &error);
  std::stringstream ss;
  ss << "\tUnfiltered Error: " << std::fixed << std::setprecision(12) << error;
  logger.info << ss.str();
  detail::savePoints("ssrlcv-initial", out->points, pipelineOutputDir());
  logger.logState("TRIANGULATE");
}

// ---------------------------------------------------------------------------------------------------------------------
// FILTERING (include/Pipeline.cuh:98-110, src/Pipeline.cu:284-352)
struct FilteringInput {
  MatchSet matchSet;
  std::vector<ptr::value<Image>> images;

  void fromCheckpoint(std::string featureGenDir, std::string featureMatchDir, int numImages) {
    images = detail::imagesFromCheckpoint(featureGenDir, numImages);
    matchSet.keyPoints = ptr::value<Unity<KeyPoint>>(detail::checkpointPath<KeyPoint>(featureMatchDir, 0));
    matchSet.matches = ptr::value<Unity<MultiMatch>>(detail::checkpointPath<MultiMatch>(featureMatchDir, 0));
  }
  void fromPreviousStage(TriangulationInput* triangulationInput) {
    this->images = triangulationInput->images;
    this->matchSet = triangulationInput->matchSet;
  }
};

struct FilteringOutput {
  ptr::value<Unity<float3>> points;
};

inline void doFiltering(FilteringInput* in, FilteringOutput* out) {
  PointCloudFactory pointCloudFactory;
  logger.logState("FILTER");
  std::stringstream ss;
  if (in->images.size() == 2) {
    float linearError;
    pointCloudFactory.linearCutoffFilter(&in->matchSet, in->images, 100.0);  // removes linear errors over 100 km
    float sigma_filter = 3.0;
    pointCloudFactory.deterministicStatisticalFilter(&in->matchSet, in->images, sigma_filter, 0.1);  // 10 % sample, 3 sigma
    BundleSet bundleSet = pointCloudFactory.generateBundles(&in->matchSet, in->images);
    out->points = pointCloudFactory.twoViewTriangulate(bundleSet, &linearError);
    ss << "Filtered " << sigma_filter << " Linear Error: " << std::fixed << std::setprecision(12) << linearError;
  } else {
    float angularError;
    pointCloudFactory.deterministicStatisticalFilter(&in->matchSet, in->images, 3.0, 0.1);
    BundleSet bundleSet = pointCloudFactory.generateBundles(&in->matchSet, in->images);
    out->points = pointCloudFactory.nViewTriangulate(bundleSet, &angularError);
    ss << "Filtered " << 0.1 << " Linear Error: " << std::fixed << std::setprecision(12) << angularError;
  }
  logger.info << ss.str();
  detail::savePoints("ssrlcv-filtered", out->points, pipelineOutputDir());
  logger.logState("FILTER");
}

// ---------------------------------------------------------------------------------------------------------------------
// BUNDLE ADJUSTMENT (include/Pipeline.cuh:116-134, src/Pipeline.cu:358-384)
struct BundleAdjustInput {
  MatchSet matchSet;
  std::vector<ptr::value<Image>> images;

  void fromCheckpoint(std::string featureGenDir, std::string filteringDir, int numImages) {
    images = detail::imagesFromCheckpoint(featureGenDir, numImages);
    matchSet.keyPoints = ptr::value<Unity<KeyPoint>>(detail::checkpointPath<KeyPoint>(filteringDir, 0));
    matchSet.matches = ptr::value<Unity<MultiMatch>>(detail::checkpointPath<MultiMatch>(filteringDir, 0));
  }
  void fromPreviousStage(FilteringInput* filteringInput) {
    this->images = filteringInput->images;
    this->matchSet = filteringInput->matchSet;
  }
};

struct BundleAdjustOutput {
  ptr::value<Unity<float3>> points;
};

inline void doBundleAdjust(BundleAdjustInput* in, BundleAdjustOutput* out) {
  if (in->images.size() != 2) return;  // not implemented for N-view in the reference either
  PointCloudFactory pointCloudFactory;
  logger.logState("BA");
  out->points = pointCloudFactory.BundleAdjustTwoView(&in->matchSet, in->images, 10, "");
  detail::savePoints("ssrlcv-BA-final", out->points, pipelineOutputDir());
  logger.logState("BA");
}

}  // namespace ssrlcv
