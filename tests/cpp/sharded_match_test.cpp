// tests/cpp/sharded_match_test.cpp -- the C++ sharded generateMatchesExhaustive (host/Distributed.hpp: RCCL called directly
// on Unity<T>::device pointers) against the single-GPU method, entry for entry, on the reference's three everest views.
// Run by tests/test_host_mirror.py under -m gpu at world size 1 (RCCL refuses two ranks on one device, and a GPU box has
// one): the communicator, both exchanges and the replicated merge all execute; only the wire is trivial.  With
// WORLD_SIZE / RANK / SSRLCV_NCCL_ID_FILE in the environment the same binary runs as one rank of several (one GPU each).
// Round 5: also the rest of the sharded flow -- nViewTriangulateSharded against PointCloudFactory::generateBundles +
// nViewTriangulate (bit for bit), selectPairBundles + evaluateCameraSetsSharded against a host-built two-view subset +
// PointCloudFactory::evaluateCameraSets -- and `bench-flow`: the whole flow through these class-level calls, timed.
//   usage: sharded_match_test <dir prepared like host_mirror_test's pipeline modes> <views>
//          sharded_match_test bench-flow <dir> <views> <iterations>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>
#include <chrono>

#include "ssrlcv.hpp"
#include "Distributed.hpp"

using namespace ssrlcv;

#define CHECK(c)                                                                  \
  do {                                                                            \
    if (!(c)) {                                                                   \
      std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

template <typename T> static std::string cp_path(const std::string& dir, int id) {
  return dir + "/" + std::to_string(id) + "_" + typeid(T).name() + ".uty";
}

// the K = 612 camera-parameter sets one BundleAdjustTwoView iteration evaluates (24 gradient points, h = 1e-5; 12 x 5
// diagonal and 132 x 4 cross points of the Hessian stencils, h = {1e-4 x3, 1e-5 x3} per camera: src/PointCloudFactory.cu:1061-1062,
// 1261,1375-1376,1444-1445), camera-major {pos.xyz, rot.xyz}: the workload ssrlcv_amd/pipeline.py ba_parameter_sets builds
static std::vector<float> ba_parameter_sets(const std::vector<ptr::value<Image>>& two) {
  float base[12];
  for (int i = 0; i < 2; ++i) {
    const Image::Camera& cam = two[(size_t)i]->camera;
    const float v[6] = {cam.cam_pos.x, cam.cam_pos.y, cam.cam_pos.z, cam.cam_rot.x, cam.cam_rot.y, cam.cam_rot.z};
    std::memcpy(base + 6 * i, v, sizeof v);
  }
  const float hlin = 1e-5f, hs[12] = {1e-4f, 1e-4f, 1e-4f, 1e-5f, 1e-5f, 1e-5f, 1e-4f, 1e-4f, 1e-4f, 1e-5f, 1e-5f, 1e-5f};
  std::vector<float> sets;
  auto push = [&](int i, float di, int j, float dj) {
    float p[12];
    std::memcpy(p, base, sizeof p);
    p[i] += di;
    if (j >= 0) p[j] += dj;
    sets.insert(sets.end(), p, p + 12);
  };
  for (int i = 0; i < 12; ++i) { push(i, hlin, -1, 0); push(i, -hlin, -1, 0); }
  for (int i = 0; i < 12; ++i)
    for (int m = -2; m <= 2; ++m) push(i, (float)m * hs[i], -1, 0);
  for (int i = 0; i < 12; ++i)
    for (int j = 0; j < 12; ++j) {
      if (i == j) continue;
      push(i, hs[i], j, hs[j]); push(i, hs[i], j, -hs[j]); push(i, -hs[i], j, hs[j]); push(i, -hs[i], j, -hs[j]);
    }
  return sets;  // 612 x 12
}

static std::vector<ptr::value<Image>> load_images(const std::string& dir, int views) {
  std::vector<ptr::value<Image>> images;
  for (int i = 0; i < views; ++i) {
    ptr::value<Image> im(dir + "/" + std::to_string(i) + "_" + typeid(Image).name() + ".cpimg", i);
    im->pixels = ptr::value<Unity<unsigned char>>(dir + "/pixels_" + std::to_string(i) + ".uty");
    images.push_back(im);
  }
  return images;
}

// bench-flow: BASELINE config[3]'s flow through the class-level calls of the mirror (SIFT_FeatureFactory::generateFeatures per
// image -> generateMatchesExhaustiveSharded -> nViewTriangulateSharded -> selectPairBundles + the 612-point sweep), world 1,
// pixels starting on the host like the reference's images; one JSON line with the stage times, the cloud as a checkpoint.
static int bench_flow(dist::Comm& comm, const std::string& dir, int views, int iters) {
  using clk = std::chrono::steady_clock;
  std::vector<ptr::value<Image>> images = load_images(dir, views);
  SIFT_FeatureFactory featureFactory(1.5f, 6.0f);
  auto seed = ptr::value<Unity<Feature<SIFT_Descriptor>>>(cp_path<Feature<SIFT_Descriptor>>(dir, -1));
  MatchFactory<SIFT_Descriptor> matchFactory(0.6f, 200.0f * 200.0f);
  matchFactory.setSeedFeatures(seed);
  if (std::getenv("SSRLCV_FLOW_DIAG")) {  // where a synchronous per-image call spends its time on this box
    auto tick = [](clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); };
    const size_t n = images[0]->pixels->size();
    ptr::device<unsigned char> d((long)n);
    for (int rep = 0; rep < 3; ++rep) {
      auto a = clk::now();
      HipSafeCall(ssrlcv_hip_memcpy(d.get(), images[0]->pixels->host.get(), n, 0));
      const double h2d = tick(a);
      a = clk::now();
      HipSafeCall(ssrlcv_hip_memcpy(images[0]->pixels->host.get(), d.get(), n, 1));
      const double d2h = tick(a);
      a = clk::now();
      auto f = featureFactory.generateFeatures(images[0], false, 2, 0.8);
      const double gen = tick(a);
      a = clk::now();
      images[0]->pixels->setMemoryState(gpu);
      const double up = tick(a);
      a = clk::now();
      auto g = featureFactory.generateFeatures(images[0], false, 2, 0.8);
      const double genDev = tick(a);
      a = clk::now();
      images[0]->pixels->setMemoryState(cpu);
      const double down = tick(a);
      std::fprintf(stderr, "diag: memcpy %zu B pageable H2D %.3f ms, D2H %.3f ms; generateFeatures from cpu-state pixels %.3f ms, from gpu-state %.3f ms; "
                   "setMemoryState(gpu) %.3f ms, (cpu) %.3f ms; %lu features\n", n, h2d, d2h, gen, genDev, up, down, f->size());
    }
  }
  double ms[5] = {0, 0, 0, 0, 0};
  unsigned long nMatches = 0, nPair = 0;
  ptr::value<Unity<float3>> cloud;
  for (int it = -1; it < iters; ++it) {  // it == -1: warm-up (plans, workspaces, communicator buffers)
    auto t0 = clk::now();
    std::vector<ptr::value<Unity<Feature<SIFT_Descriptor>>>> feats((size_t)views);
    for (int v = 0; v < views; ++v)
      if (dist::imageOwner(v, comm.world) == comm.rank) feats[(size_t)v] = featureFactory.generateFeatures(images[(size_t)v], false, 2, 0.8);
    auto t1 = clk::now();
    MatchSet matchSet = dist::generateMatchesExhaustiveSharded(comm, matchFactory, images, feats, 25.0f, 5.0f);
    auto t2 = clk::now();
    cloud = dist::nViewTriangulateSharded(comm, &matchSet, images);
    auto t3 = clk::now();
    std::vector<ptr::value<Image>> two = {images[0], images[1]};
    MatchSet pairSet = dist::selectPairBundles(&matchSet, 0, 1);
    const std::vector<float> params = ba_parameter_sets(two);
    std::vector<float> sums = dist::evaluateCameraSetsSharded(comm, &pairSet, two, params, 612);
    HipSafeCall(ssrlcv_hip_device_synchronize());
    auto t4 = clk::now();
    if (it >= 0) {
      auto d = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
      ms[0] += d(t0, t1); ms[1] += d(t1, t2); ms[2] += d(t2, t3); ms[3] += d(t3, t4); ms[4] += d(t0, t4);
    }
    nMatches = matchSet.matches->size();
    nPair = pairSet.matches != nullptr ? pairSet.matches->size() : 0;
    if (!std::isfinite(sums[0])) return 1;
  }
  if (comm.rank == 0) {
    cloud->checkpoint(300, dir + "/");
    std::printf("{\"flow_ms\": %.3f, \"sift_ms\": %.3f, \"match_merge_ms\": %.3f, \"triangulate_ms\": %.3f, \"ba_sweep_ms\": %.3f, "
                "\"multi_matches\": %lu, \"points\": %lu, \"ba_bundles\": %lu, \"views\": %d, \"world\": %d}\n",
                ms[4] / iters, ms[0] / iters, ms[1] / iters, ms[2] / iters, ms[3] / iters, nMatches, cloud->size(), nPair, views, comm.world);
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s <dir> <views> | bench-flow <dir> <views> <iterations>\n", argv[0]); return 2; }
  const bool flow = std::string(argv[1]) == "bench-flow";
  if (flow && argc < 5) { std::fprintf(stderr, "usage: %s bench-flow <dir> <views> <iterations>\n", argv[0]); return 2; }
  const std::string dir = flow ? argv[2] : argv[1];
  const int views = std::atoi(flow ? argv[3] : argv[2]);
  const int world = std::getenv("WORLD_SIZE") ? std::atoi(std::getenv("WORLD_SIZE")) : 1;
  const int rank = std::getenv("RANK") ? std::atoi(std::getenv("RANK")) : 0;
  try {
    dist::hipCheck(hipSetDevice(std::getenv("LOCAL_RANK") ? std::atoi(std::getenv("LOCAL_RANK")) : 0), "hipSetDevice");
    // rendezvous: rank 0 writes the RCCL id to a file the others poll (a node-local launch needs nothing more)
    ncclUniqueId id;
    const char* idFile = std::getenv("SSRLCV_NCCL_ID_FILE");
    if (rank == 0) {
      dist::ncclCheck(ncclGetUniqueId(&id), "ncclGetUniqueId");
      if (world > 1) {
        if (!idFile) { std::fprintf(stderr, "SSRLCV_NCCL_ID_FILE must be set for WORLD_SIZE > 1\n"); return 2; }
        std::ofstream(std::string(idFile) + ".tmp", std::ios::binary).write(reinterpret_cast<const char*>(&id), sizeof id);
        std::rename((std::string(idFile) + ".tmp").c_str(), idFile);
      }
    } else {
      if (!idFile) return 2;
      for (int tries = 0; tries < 600; ++tries) {
        std::ifstream f(idFile, std::ios::binary);
        if (f && f.read(reinterpret_cast<char*>(&id), sizeof id)) break;
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
      }
    }
    dist::Comm comm{nullptr, rank, world, nullptr};
    dist::ncclCheck(ncclCommInitRank(&comm.comm, world, id, rank), "ncclCommInitRank");
    if (flow) {
      const int rc = bench_flow(comm, dir, views, std::atoi(argv[4]));
      dist::ncclCheck(ncclCommDestroy(comm.comm), "ncclCommDestroy");
      return rc;
    }

    std::vector<ptr::value<Image>> images = load_images(dir, views);
    SIFT_FeatureFactory featureFactory(1.5f, 6.0f);
    auto seed = ptr::value<Unity<Feature<SIFT_Descriptor>>>(cp_path<Feature<SIFT_Descriptor>>(dir, -1));
    // every rank extracts the images it owns; the reference result (every image, every pair, one GPU) on rank 0
    std::vector<ptr::value<Unity<Feature<SIFT_Descriptor>>>> mine((size_t)views), all((size_t)views);
    for (int v = 0; v < views; ++v) {
      if (dist::imageOwner(v, world) == rank) mine[(size_t)v] = featureFactory.generateFeatures(images[(size_t)v], false, 2, 0.8);
      if (rank == 0) all[(size_t)v] = featureFactory.generateFeatures(images[(size_t)v], false, 2, 0.8);
    }
    MatchFactory<SIFT_Descriptor> shardedFactory(0.6f, 200.0f * 200.0f);
    shardedFactory.setSeedFeatures(seed);
    MatchSet sharded = dist::generateMatchesExhaustiveSharded(comm, shardedFactory, images, mine, 25.0f, 5.0f);
    sharded.matches->transferMemoryTo(cpu);
    sharded.keyPoints->transferMemoryTo(cpu);
    std::printf("rank %d of %d: sharded %lu multi-matches, %lu key points\n", rank, world, sharded.matches->size(), sharded.keyPoints->size());
    if (rank == 0) {
      MatchFactory<SIFT_Descriptor> singleFactory(0.6f, 200.0f * 200.0f);
      singleFactory.setSeedFeatures(seed);
      MatchSet single = singleFactory.generateMatchesExhaustive(images, all, 25.0f, 5.0f);
      single.matches->transferMemoryTo(cpu);
      single.keyPoints->transferMemoryTo(cpu);
      CHECK(single.matches->size() == sharded.matches->size() && single.keyPoints->size() == sharded.keyPoints->size());
      CHECK(std::memcmp(single.matches->host.get(), sharded.matches->host.get(), sizeof(MultiMatch) * single.matches->size()) == 0);
      for (unsigned long k = 0; k < single.keyPoints->size(); ++k) {
        const KeyPoint &a = single.keyPoints->host.get()[k], &b = sharded.keyPoints->host.get()[k];
        CHECK(a.parentId == b.parentId && a.loc.x == b.loc.x && a.loc.y == b.loc.y);
      }
      // the received feature arrays are the owners'
      for (int v = 0; v < views; ++v) {
        mine[(size_t)v]->transferMemoryTo(cpu);
        all[(size_t)v]->transferMemoryTo(cpu);
        CHECK(mine[(size_t)v]->size() == all[(size_t)v]->size());
        CHECK(std::memcmp(mine[(size_t)v]->host.get(), all[(size_t)v]->host.get(), sizeof(Feature<SIFT_Descriptor>) * all[(size_t)v]->size()) == 0);
      }
      sharded.keyPoints->checkpoint(200, dir + "/");
      sharded.matches->checkpoint(200, dir + "/");
      std::printf("sharded == single: %lu multi-matches\n", single.matches->size());
    }
    // ---- stage C and the BA sweep, sharded, against the single-GPU class methods
    {
      PointCloudFactory pcf;
      ptr::value<Unity<float3>> cloud = dist::nViewTriangulateSharded(comm, &sharded, images);
      CHECK(cloud->getMemoryState() == cpu && sharded.matches->getMemoryState() == cpu);
      std::vector<ptr::value<Image>> two = {images[0], images[1]};
      MatchSet pairSet = dist::selectPairBundles(&sharded, 0, 1);
      CHECK(pairSet.matches != nullptr && pairSet.matches->getMemoryState() == gpu);
      const std::vector<float> params = ba_parameter_sets(two);
      CHECK(params.size() == 612u * 12u);
      std::vector<float> sums = dist::evaluateCameraSetsSharded(comm, &pairSet, two, params, 612);
      if (rank == 0) {
        BundleSet bs = pcf.generateBundles(&sharded, images);
        ptr::value<Unity<float3>> ref = views > 2 ? pcf.nViewTriangulate(bs) : pcf.twoViewTriangulate(bs);
        CHECK(ref->size() == cloud->size());
        CHECK(std::memcmp(ref->host.get(), cloud->host.get(), sizeof(float3) * ref->size()) == 0);
        // the pair's bundles selected on the host, the way the flow was written before the library pass existed
        std::vector<KeyPoint> kps;
        for (unsigned long m = 0; m < sharded.matches->size(); ++m) {
          const MultiMatch mm = sharded.matches->host.get()[m];
          const KeyPoint* kp = sharded.keyPoints->host.get() + mm.index;
          if (mm.numKeyPoints == 2 && kp[0].parentId == 0 && kp[1].parentId == 1) { kps.push_back(kp[0]); kps.push_back(kp[1]); }
        }
        CHECK(kps.size() == 2 * pairSet.matches->size() && !kps.empty());
        pairSet.matches->transferMemoryTo(cpu);
        pairSet.keyPoints->transferMemoryTo(cpu);
        MatchSet hostSet;
        hostSet.matches = ptr::value<Unity<MultiMatch>>(nullptr, (unsigned long)(kps.size() / 2), cpu);
        hostSet.keyPoints = ptr::value<Unity<KeyPoint>>(nullptr, (unsigned long)kps.size(), cpu);
        for (unsigned long j = 0; j < kps.size() / 2; ++j) {
          hostSet.matches->host.get()[j] = MultiMatch{2u, (int)(2 * j)};
          KeyPoint a = kps[2 * j], b = kps[2 * j + 1];
          a.parentId = 0;
          b.parentId = 1;
          hostSet.keyPoints->host.get()[2 * j] = a;
          hostSet.keyPoints->host.get()[2 * j + 1] = b;
          const MultiMatch got = pairSet.matches->host.get()[j];
          CHECK(got.numKeyPoints == 2u && got.index == (int)(2 * j));
          for (int e = 0; e < 2; ++e) {
            const KeyPoint &g = pairSet.keyPoints->host.get()[2 * j + e], &w = hostSet.keyPoints->host.get()[2 * j + e];
            CHECK(g.parentId == w.parentId && g.loc.x == w.loc.x && g.loc.y == w.loc.y);
          }
        }
        if (world == 1) {  // one rank: the sweep's sums are the single-GPU sums bit for bit (the same launch over the same bundles)
          std::vector<float> want = pcf.evaluateCameraSets(&hostSet, two, params, 612);
          // (the launch adds its blocks' partial sums with float atomics: the order, hence the last bits, vary from run to run)
          for (int k = 0; k < 612; ++k) CHECK(std::fabs(want[(size_t)k] - sums[(size_t)k]) <= 1e-4f * std::fabs(want[(size_t)k]));
        }
        cloud->checkpoint(200, dir + "/");
        std::printf("sharded cloud == single: %lu points; pair (0,1): %lu bundles, f(base) = %g\n", cloud->size(), pairSet.matches->size(), sums[24 + 2]);
      }
    }
    dist::ncclCheck(ncclCommDestroy(comm.comm), "ncclCommDestroy");
  } catch (std::exception& e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 3;
  }
  std::printf("sharded ok\n");
  return 0;
}
