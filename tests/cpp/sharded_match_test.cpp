// tests/cpp/sharded_match_test.cpp -- the C++ sharded generateMatchesExhaustive (host/Distributed.hpp: RCCL called directly
// on Unity<T>::device pointers) against the single-GPU method, entry for entry, on the reference's three everest views.
// Run by tests/test_host_mirror.py under -m gpu at world size 1 (RCCL refuses two ranks on one device, and a GPU box has
// one): the communicator, both exchanges and the replicated merge all execute; only the wire is trivial.  With
// WORLD_SIZE / RANK / SSRLCV_NCCL_ID_FILE in the environment the same binary runs as one rank of several (one GPU each).
//   usage: sharded_match_test <dir prepared like host_mirror_test's pipeline modes> <views>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
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

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s <dir> <views>\n", argv[0]); return 2; }
  const std::string dir = argv[1];
  const int views = std::atoi(argv[2]);
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

    std::vector<ptr::value<Image>> images;
    for (int i = 0; i < views; ++i) {
      ptr::value<Image> im(dir + "/" + std::to_string(i) + "_" + typeid(Image).name() + ".cpimg", i);
      im->pixels = ptr::value<Unity<unsigned char>>(dir + "/pixels_" + std::to_string(i) + ".uty");
      images.push_back(im);
    }
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
    dist::ncclCheck(ncclCommDestroy(comm.comm), "ncclCommDestroy");
  } catch (std::exception& e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 3;
  }
  std::printf("sharded ok\n");
  return 0;
}
