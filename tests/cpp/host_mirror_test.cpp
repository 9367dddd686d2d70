// tests/cpp/host_mirror_test.cpp -- exercises the C++ host mirror of the reference API (ssrlcv_amd/host/*.hpp).
//   typeinfo            print typeid name / hash_code of the checkpointable types (compared with the reference's .uty headers)
//   cpu <dir>           Unity<T> state machine, exceptions, checkpoint round trip, Image .cpimg reader (no GPU needed)
//   pipeline2 <dir>     (GPU) SIFT -> seed distances -> double-constrained match -> MatchSet -> triangulate -> BA, through
//                       the reference's class API; inputs/outputs are .uty / .cpimg files in <dir>
//   pipeline3 <dir>     (GPU) 3-view: generateMatchesExhaustive -> nViewTriangulate
//   pinv <in> <out>     pseudoInverse of every 12 x 12 float matrix in a file (no GPU needed)
//   bench <raw> W H n   (GPU) times SIFT_FeatureFactory::generateFeatures from host-state pixels (PCIe included)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include "ssrlcv.hpp"

using namespace ssrlcv;

#define CHECK(cond)                                                                  \
  do {                                                                               \
    if (!(cond)) { std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
  } while (0)

template <typename T> static std::string cp_path(const std::string& dir, int id) {
  return dir + "/" + std::to_string(id) + "_" + typeid(T).name() + ".uty";
}

static int typeinfo_mode() {
  auto p = [](const char* label, const std::type_info& t) { std::printf("%s %s %zu\n", label, t.name(), t.hash_code()); };
  p("uchar", typeid(unsigned char));
  p("float", typeid(float));
  p("float3", typeid(float3));
  p("KeyPoint", typeid(KeyPoint));
  p("MultiMatch", typeid(MultiMatch));
  p("Feature", typeid(Feature<SIFT_Descriptor>));
  p("Image", typeid(Image));
  return 0;
}

static bool is_even(const float& v) { return ((int)v) % 2 == 0; }
static bool kp_less(const KeyPoint& a, const KeyPoint& b) { return a.parentId < b.parentId; }

static int cpu_mode(const std::string& dir) {
  // construction, ownership, resize, clear
  ptr::host<float> h(8);
  for (int i = 0; i < 8; ++i) h.get()[i] = (float)i;
  Unity<float> u(h, 8, cpu);
  CHECK(u.size() == 8 && u.getMemoryState() == cpu && u.getFore() == cpu && u.host.get() == h.get());
  u.resize(5);
  CHECK(u.size() == 5 && u.host.get()[4] == 4.0f);
  u.remove(is_even);
  CHECK(u.size() == 2 && u.host.get()[0] == 1.0f && u.host.get()[1] == 3.0f);
  // exceptions of the state machine (include/Unity.cuh:77-133)
  bool threw = false;
  try { Unity<float> n; n.transferMemoryTo(gpu); } catch (NullUnityException&) { threw = true; }
  CHECK(threw);
  threw = false;
  try { Unity<float> n; n.setData(ptr::host<float>(2), 2, gpu); } catch (IllegalUnityTransition&) { threw = true; }
  CHECK(threw);
  threw = false;
  try { Unity<float> n; n.setData(nullptr, 0, cpu); } catch (IllegalUnityTransition&) { threw = true; }
  CHECK(threw);
  threw = false;
  try { u.setFore(gpu); } catch (IllegalUnityTransition&) { threw = true; }
  CHECK(threw);
  threw = false;
  try { Unity<KeyPoint> bad(dir + "/missing.uty"); } catch (CheckpointException&) { threw = true; }
  CHECK(threw);
  // sort (stable) + checkpoint round trip in the reference's on-disk format
  ptr::value<Unity<KeyPoint>> kps(nullptr, 6, cpu);
  for (int i = 0; i < 6; ++i) kps->host.get()[i] = {5 - i / 2, {(float)i, (float)(10 * i)}};
  kps->sort(kp_less);
  CHECK(kps->host.get()[0].parentId == 3 && kps->host.get()[0].loc.x == 4.0f && kps->host.get()[1].loc.x == 5.0f);
  kps->checkpoint(7, dir + "/");
  Unity<KeyPoint> back(cp_path<KeyPoint>(dir, 7));
  CHECK(back.size() == 6 && back.getMemoryState() == cpu);
  CHECK(std::memcmp(back.host.get(), kps->host.get(), 6 * sizeof(KeyPoint)) == 0);
  threw = false;
  try { Unity<MultiMatch> wrong(cp_path<KeyPoint>(dir, 7)); } catch (CheckpointException&) { threw = true; }
  CHECK(threw);
  // files written by the Python side with the REFERENCE's header bytes must load: proves typeid compatibility
  Unity<float3> pts(cp_path<float3>(dir, 0));
  CHECK(pts.size() == 3 && pts.host.get()[2].z == 9.0f);
  Image img(dir + "/0_" + typeid(Image).name() + ".cpimg", 0);
  CHECK(img.id == 0 && img.size.x == 1024 && img.colorDepth == 1 && img.camera.foc > 0.85f && img.camera.foc < 0.87f);
  // ASCII PLY writer (src/io_util.cpp:740-754)
  {
    ptr::value<Unity<float3>> cloud(nullptr, 2, cpu);
    cloud->host.get()[0] = {1.5f, -2.0f, 3.25f};
    cloud->host.get()[1] = {0.0f, 1e-3f, 400.125f};
    writePLY("cloud", cloud, dir + "/");
    std::ifstream in(dir + "/cloud.ply");
    std::string all((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    CHECK(all.find("element vertex 2\n") != std::string::npos);
    CHECK(all.find("end_header\n1.5 -2 3.25\n0 0.001 400.125\n") != std::string::npos);
  }
  // Feature() default state that Unity(nullptr, n, gpu) relies on
  Feature<SIFT_Descriptor> f;
  CHECK(f.parent == -1 && f.loc.x == -1.0f && f.descriptor.theta == 0.0f);
  std::printf("cpu ok\n");
  return 0;
}

static std::vector<ptr::value<Image>> load_images(const std::string& dir, int n) {
  std::vector<ptr::value<Image>> images;
  for (int i = 0; i < n; ++i) {
    ptr::value<Image> im(dir + "/" + std::to_string(i) + "_" + typeid(Image).name() + ".cpimg", i);
    im->pixels = ptr::value<Unity<unsigned char>>(dir + "/pixels_" + std::to_string(i) + ".uty");
    images.push_back(im);
  }
  return images;
}

static int pipeline_mode(const std::string& dir, int views) {
  std::vector<ptr::value<Image>> images = load_images(dir, views);
  SIFT_FeatureFactory featureFactory(1.5f, 6.0f);  // src/Pipeline.cu:17
  std::vector<ptr::value<Unity<Feature<SIFT_Descriptor>>>> allFeatures;
  for (auto& im : images) {
    auto feats = featureFactory.generateFeatures(im, false, 2, 0.8);
    CHECK(feats->getMemoryState() == gpu && im->pixels->getMemoryState() == cpu);  // pixels restored to origin
    feats->transferMemoryTo(cpu);  // src/Pipeline.cu:45
    allFeatures.push_back(feats);
    std::printf("features %d %lu\n", im->id, feats->size());
  }
  // the reference's stage flow (src/SFM.cu:131-230, test/Pipeline.cu) through the Pipeline.hpp glue
  pipelineOutputDir() = dir + "/";
  PoseEstimationInput poseInput;
  poseInput.seedFeatures = ptr::value<Unity<Feature<SIFT_Descriptor>>>(cp_path<Feature<SIFT_Descriptor>>(dir, -1));
  poseInput.allFeatures = allFeatures;
  poseInput.images = images;
  PoseEstimationOutput poseOutput;  // pose estimation skipped, as `SFM` does without --pose
  FeatureMatchingInput matchInput;
  matchInput.fromPreviousStage(&poseInput, &poseOutput, 25.0f, 5.0f);
  FeatureMatchingOutput matchOutput;
  doFeatureMatching(&matchInput, &matchOutput);
  if (views == 2) CHECK(allFeatures[0]->getMemoryState() == both);  // origin state restored by the match factory
  MatchSet& matchSet = matchOutput.matchSet;
  matchSet.keyPoints->checkpoint(100, dir + "/");
  matchSet.matches->checkpoint(100, dir + "/");
  TriangulationInput triInput;
  triInput.fromPreviousStage(&matchInput, &matchOutput);
  TriangulationOutput triOutput;
  doTriangulation(&triInput, &triOutput);
  CHECK(triInput.matchSet.matches->getMemoryState() == cpu);
  ptr::value<Unity<float3>> points = triOutput.points;
  CHECK(points->getMemoryState() == cpu);
  points->checkpoint(100, dir + "/");
  std::printf("matches %lu\n", matchSet.matches->size());
  PointCloudFactory pcf;
  if (views == 2) {
    // doBundleAdjust (src/Pipeline.cu:371-384): 10 iterations requested
    FilteringInput unfiltered;  // BA on the unfiltered set here; the filters have their own modes below
    unfiltered.fromPreviousStage(&triInput);
    BundleAdjustInput baInput;
    baInput.fromPreviousStage(&unfiltered);
    BundleAdjustOutput baOutput;
    doBundleAdjust(&baInput, &baOutput);
    auto adjusted = baOutput.points;
    CHECK(adjusted != nullptr && adjusted->size() == points->size());
    adjusted->checkpoint(101, dir + "/");
    // pseudo-inverse self check: H H+ H ~ H on a symmetric rank-deficient matrix
    ptr::value<Unity<float>> H(nullptr, 144, cpu);
    for (int i = 0; i < 12; ++i)
      for (int j = 0; j < 12; ++j) H->host.get()[i * 12 + j] = (i < 9 && j < 9) ? (float)((i + 1) * (j + 1) % 7) + (i == j ? 20.0f : 0.0f) : 0.0f;
    for (int i = 0; i < 12; ++i)
      for (int j = 0; j < i; ++j) H->host.get()[i * 12 + j] = H->host.get()[j * 12 + i];
    auto Hp = pcf.calculateImageHessianInverse(H);
    double worst = 0;
    for (int i = 0; i < 12; ++i)
      for (int j = 0; j < 12; ++j) {
        double acc = 0;
        for (int a = 0; a < 12; ++a)
          for (int b = 0; b < 12; ++b) acc += (double)H->host.get()[i * 12 + a] * Hp->host.get()[a * 12 + b] * H->host.get()[b * 12 + j];
        worst = std::max(worst, std::fabs(acc - H->host.get()[i * 12 + j]));
      }
    std::printf("pinv residual %g\n", worst);
    CHECK(worst < 1e-3);
  }
  std::printf("pipeline ok\n");
  return 0;
}

// doFiltering (src/Pipeline.cu:297-352) from the reference's stage-0 MatchSet checkpoint
static int filter_mode(const std::string& dir, int views) {
  pipelineOutputDir() = dir + "/";
  FilteringInput in;
  in.fromCheckpoint(dir, dir, views);  // <dir>/i_Image.cpimg + 0_KeyPoint / 0_MultiMatch, like test/Pipeline.cu
  FilteringOutput out;
  doFiltering(&in, &out);
  in.matchSet.keyPoints->checkpoint(201, dir + "/");
  in.matchSet.matches->checkpoint(201, dir + "/");
  out.points->checkpoint(201, dir + "/");
  std::printf("filtered %lu\nfilter ok\n", in.matchSet.matches->size());
  return 0;
}

// PoseEstimator::LM_optimize on the reference's stage-0 MatchSet (pairs) of the 2-view fixture: started from the true
// relative pose the cost must not grow; started from a target camera rotated by 2 mrad it must come back down
static int pose_mode(const std::string& dir) {
  TriangulationInput in;
  in.fromCheckpoint(dir, dir, 2);
  unsigned long n = in.matchSet.matches->size();
  ptr::value<Unity<Match>> matches(nullptr, n, cpu);
  for (unsigned long i = 0; i < n; ++i) {
    Match m;
    m.invalid = false;
    m.keyPoints[0] = in.matchSet.keyPoints->host.get()[2 * i];
    m.keyPoints[1] = in.matchSet.keyPoints->host.get()[2 * i + 1];
    matches->host.get()[i] = m;
  }
  auto relative = [&](Pose& pose) {
    float C0[3][3], C0t[3][3], C1[3][3], rel[3][3];
    getRotationMatrix(in.images[0]->camera.cam_rot, C0);
    transpose(C0, C0t);
    getRotationMatrix(in.images[1]->camera.cam_rot, C1);
    multiply(C0t, C1, rel);
    float3 rot = getAxisRotations(rel);
    pose.roll = rot.x; pose.pitch = rot.y; pose.yaw = rot.z;
    pose.x = pose.y = pose.z = 0;  // LM_optimize fills the position from the cameras
  };
  PoseEstimator estim(in.images[0], in.images[1], matches);
  Pose pose;
  relative(pose);
  Pose start = pose;
  float lambda = 100;
  estim.LM_optimize(&pose);  // sets x, y, z and iterates
  std::printf("pose true-start roll %g pitch %g yaw %g -> %g %g %g\n", start.roll, start.pitch, start.yaw, pose.roll, pose.pitch, pose.yaw);
  CHECK(std::isfinite(pose.roll) && std::isfinite(pose.pitch) && std::isfinite(pose.yaw));
  CHECK(std::fabs(pose.roll - start.roll) < 5e-3f && std::fabs(pose.pitch - start.pitch) < 5e-3f && std::fabs(pose.yaw - start.yaw) < 5e-3f);
  // perturbed start: one LM_iteration at a time, costs reported through the terms call
  Pose bad = pose;
  bad.pitch += 0.002f;
  Pose it = bad;
  lambda = 100;
  int iters = 0;
  while (iters < 50 && estim.LM_iteration(&it, &lambda)) ++iters;
  std::printf("pose perturbed pitch %g -> %g (refined %g) after %d iterations\n", bad.pitch, it.pitch, pose.pitch, iters);
  CHECK(iters >= 1);
  CHECK(std::fabs(it.pitch - pose.pitch) < 0.5f * std::fabs(bad.pitch - pose.pitch));  // moved back towards the optimum
  CHECK(matches->getMemoryState() == cpu);  // origin state restored
  std::printf("pose ok\n");
  return 0;
}

// bench <raw u8 file> <W> <H> <iters>: the drop-in class path as a reference caller sees it.  Pixels start in host
// memory (Unity state cpu, pinned) for every call, so each generateFeatures includes the H2D copy of the image; the
// second figure also brings the features back (transferMemoryTo(cpu)), i.e. both PCIe legs.  One JSON line.
#include <chrono>
static int bench_mode(const std::string& path, unsigned W, unsigned H, int iters) {
  std::FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) { std::fprintf(stderr, "cannot open %s\n", path.c_str()); return 2; }
  ptr::host<unsigned char> px((long)W * H, true);
  size_t got = std::fread(px.get(), 1, (size_t)W * H, f);
  std::fclose(f);
  CHECK(got == (size_t)W * H);
  SIFT_FeatureFactory factory(1.5f, 6.0f);
  auto run = [&](bool fetch, double& ms, unsigned long& nfeat) {
    double total = 0;
    for (int i = -1; i < iters; ++i) {  // i == -1: warm-up (plan + workspace creation)
      ptr::value<Unity<unsigned char>> pixels(px, (unsigned long)W * H, cpu, true);
      ptr::value<Image> image(uint2{W, H}, 1u, pixels);
      image->id = 0;
      auto t0 = std::chrono::steady_clock::now();
      auto feats = factory.generateFeatures(image, false, 2, 0.8f);
      if (fetch) feats->transferMemoryTo(cpu);
      HipSafeCall(ssrlcv_hip_device_synchronize());
      auto t1 = std::chrono::steady_clock::now();
      if (i >= 0) total += std::chrono::duration<double, std::milli>(t1 - t0).count();
      nfeat = feats->size();
    }
    ms = total / iters;
  };
  double msGen = 0, msFetch = 0;
  unsigned long n = 0;
  run(false, msGen, n);
  run(true, msFetch, n);
  std::printf("{\"features\": %lu, \"ms_generateFeatures_from_host_pixels\": %.3f, \"ms_with_features_to_host\": %.3f}\n", n,
              msGen, msFetch);
  return 0;
}

// pinv <in.bin> <out.bin>: ssrlcv::pseudoInverse (host/matrix_util.hpp, the P5 replacement of cuSOLVER + cuBLAS) on every
// 12 x 12 float matrix of the input file
static int pinv_mode(const std::string& in, const std::string& out) {
  std::FILE* f = std::fopen(in.c_str(), "rb");
  CHECK(f != nullptr);
  std::vector<float> buf;
  float m[144];
  std::FILE* g = std::fopen(out.c_str(), "wb");
  CHECK(g != nullptr);
  while (std::fread(m, sizeof(float), 144, f) == 144) {
    std::vector<float> inv = pseudoInverse(m, 12);
    std::fwrite(inv.data(), sizeof(float), 144, g);
  }
  std::fclose(f);
  std::fclose(g);
  return 0;
}

int main(int argc, char** argv) {
  std::string mode = argc > 1 ? argv[1] : "typeinfo";
  try {
    if (mode == "typeinfo") return typeinfo_mode();
    if (argc < 3) { std::fprintf(stderr, "usage: %s <mode> <dir>\n", argv[0]); return 2; }
    if (mode == "bench") {
      if (argc < 6) { std::fprintf(stderr, "usage: %s bench <raw u8 file> <W> <H> <iters>\n", argv[0]); return 2; }
      return bench_mode(argv[2], (unsigned)std::atoi(argv[3]), (unsigned)std::atoi(argv[4]), std::atoi(argv[5]));
    }
    if (mode == "pinv") {
      if (argc < 4) { std::fprintf(stderr, "usage: %s pinv <in.bin> <out.bin>\n", argv[0]); return 2; }
      return pinv_mode(argv[2], argv[3]);
    }
    if (mode == "cpu") return cpu_mode(argv[2]);
    if (mode == "pipeline2") return pipeline_mode(argv[2], 2);
    if (mode == "pipeline3") return pipeline_mode(argv[2], 3);
    if (mode == "filter2") return filter_mode(argv[2], 2);
    if (mode == "filter3") return filter_mode(argv[2], 3);
    if (mode == "pose") return pose_mode(argv[2]);
  } catch (std::exception& e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 3;
  }
  return 2;
}
