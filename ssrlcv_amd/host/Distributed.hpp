// ssrlcv_amd/host/Distributed.hpp -- MatchFactory<T>::generateMatchesExhaustive sharded over the GPUs of one node
// (BASELINE.json config[3]; reference: src/MatchFactory.cu:907-1028, which runs every pair on one GPU).
//
// One process per GPU; the communicator is RCCL's (ncclComm_t over xGMI), called directly on Unity<T>::device pointers.
// Units are independent -- SIFT per image (image v belongs to rank v mod G), matching per image pair -- so the flow has
// exactly two exchanges:
//   1. every rank receives every image's Feature array (ncclBroadcast from its owner, all images in one group), after an
//      all-reduce of the feature counts;
//   2. the validated uint2_pair array of every pair goes from the rank that matched it to everybody (same pattern), after
//      an all-reduce of the pair counts;
// then the merge + KeyPoint table run replicated (MatchFactory<T>::assembleMatchSet: deterministic, so every rank ends
// with the same MatchSet).  Pairs go to ranks by the library's longest-processing-time-first table
// (ssrlcv_assign_pairs_host: cost nq x nt, the same definition the Python driver ssrlcv_amd/dist.py uses).
// Exact sizes travel: no padding to the largest rank.
//
// Not part of the umbrella header ssrlcv.hpp: a program that includes it links librccl and libamdhip64 (still plain g++:
// rccl_abi.hpp declares the few entry points used); everything else of the mirror needs only the C ABI library.
#pragma once
#include "rccl_abi.hpp"

#include <stdexcept>
#include <string>
#include <vector>

#include "MatchFactory.hpp"

namespace ssrlcv {
namespace dist {

inline void ncclCheck(ncclResult_t r, const char* what) {
  if (r != ncclSuccess) throw std::runtime_error(std::string("RCCL: ") + what + ": " + ncclGetErrorString(r));
}
inline void hipCheck(int e, const char* what) {
  if (e != 0) throw std::runtime_error(std::string("HIP: ") + what + ": " + hipGetErrorString(e));
}

struct Comm {
  ncclComm_t comm;
  int rank, world;
  hipStream_t stream;  // the exchanges' stream (nullptr = the default stream, which the mirror's synchronous calls use)
};

inline int imageOwner(int image, int world) { return image % world; }

// owners[p] for the pairs (0,1),(0,2)..(1,2).. -- the library's table
inline std::vector<uint32_t> assignPairs(const std::vector<uint32_t>& numFeatures, int world) {
  const size_t V = numFeatures.size();
  std::vector<uint32_t> owners(V * (V - 1) / 2);
  if (V >= 2) HipSafeCall(ssrlcv_assign_pairs_host((uint32_t)V, numFeatures.data(), (uint32_t)world, owners.data()));
  return owners;
}

// in-place sum of a small host vector over the ranks (counts): H2D, ncclAllReduce, D2H
inline void allReduceCounts(Comm& c, std::vector<uint32_t>& v) {
  if (v.empty()) return;
  ptr::device<uint32_t> d((long)v.size());
  HipSafeCall(ssrlcv_hip_memcpy(d.get(), v.data(), v.size() * sizeof(uint32_t), 0));
  ncclCheck(ncclAllReduce(d.get(), d.get(), v.size(), ncclUint32, ncclSum, c.comm, c.stream), "all-reduce of counts");
  hipCheck(hipStreamSynchronize(c.stream), "counts");
  HipSafeCall(ssrlcv_hip_memcpy(v.data(), d.get(), v.size() * sizeof(uint32_t), 1));
}

// Sharded generateMatchesExhaustive.  features[v] is this rank's own result for the images it owns (v mod world == rank)
// and may be null for the others: they are received.  On return every rank holds every image's features (gpu state) in
// `features` and the same MatchSet.  epsilon / delta as upstream (GEO_ORBIT == 1: the double-constrained matcher).
template <typename T>
MatchSet generateMatchesExhaustiveSharded(Comm& c, MatchFactory<T>& matchFactory, std::vector<ptr::value<Image>> images,
                                          std::vector<ptr::value<Unity<Feature<T>>>>& features, float epsilon, float delta) {
  const int V = (int)images.size();
  if ((int)features.size() != V) throw std::invalid_argument("generateMatchesExhaustiveSharded: one feature slot per image");
  // ---- exchange 1: feature arrays
  std::vector<uint32_t> nf((size_t)V, 0u);
  for (int v = 0; v < V; ++v)
    if (imageOwner(v, c.world) == c.rank) {
      if (features[(size_t)v] == nullptr) throw std::invalid_argument("generateMatchesExhaustiveSharded: an owned image has no features");
      nf[(size_t)v] = (uint32_t)features[(size_t)v]->size();
    }
  allReduceCounts(c, nf);
  for (int v = 0; v < V; ++v) {
    auto& f = features[(size_t)v];
    if (imageOwner(v, c.world) == c.rank) {
      if (f->getMemoryState() != gpu && f->getMemoryState() != both) f->setMemoryState(gpu);
      else if (f->getMemoryState() == both && f->getFore() == cpu) f->transferMemoryTo(gpu);
    } else {
      // (a Unity<T> cannot be empty, upstream's neither: an image without a single feature stops the flow there too)
      if (nf[(size_t)v] == 0) throw std::runtime_error("generateMatchesExhaustiveSharded: image " + std::to_string(v) + " has no features");
      f = ptr::value<Unity<Feature<T>>>(nullptr, (unsigned long)nf[(size_t)v], gpu);
    }
  }
  ncclCheck(ncclGroupStart(), "group");
  for (int v = 0; v < V; ++v)
    if (nf[(size_t)v])
      ncclCheck(ncclBroadcast(features[(size_t)v]->device.get(), features[(size_t)v]->device.get(), (size_t)nf[(size_t)v] * sizeof(Feature<T>),
                              ncclChar, imageOwner(v, c.world), c.comm, c.stream), "broadcast of a feature array");
  ncclCheck(ncclGroupEnd(), "group");
  hipCheck(hipStreamSynchronize(c.stream), "feature exchange");
  // ---- this rank's pairs
  const std::vector<uint32_t> owners = assignPairs(nf, c.world);
  std::vector<ptr::value<Unity<uint2_pair>>> matchIndices(owners.size());
  std::vector<uint32_t> pairCounts(owners.size(), 0u);
  size_t p = 0;
  for (int q = 0; q + 1 < V; ++q) {
    ptr::value<Unity<float>> seedDistances;  // per query image, and only if this rank matches one of its pairs
    for (int t = q + 1; t < V; ++t, ++p) {
      if ((int)owners[p] != c.rank) continue;
      if (matchFactory.hasSeedFeatures() && seedDistances == nullptr) seedDistances = matchFactory.getSeedDistances(features[(size_t)q]);
      matchIndices[p] = matchFactory.generateMatchesDoubleConstrainedIndexOnly(images[(size_t)q], features[(size_t)q], images[(size_t)t],
                                                                              features[(size_t)t], epsilon, delta, seedDistances);
      pairCounts[p] = (uint32_t)matchIndices[p]->size();
    }
  }
  // ---- exchange 2: validated pair arrays
  allReduceCounts(c, pairCounts);
  unsigned long long totalMatches = 0;
  for (size_t k = 0; k < owners.size(); ++k) {
    totalMatches += pairCounts[k];
    // (never 0: a pair without a valid match keeps its nq unvalidated entries, like upstream -- validateMatches clears a
    // copy of the pointer, src/MatchFactory.cu:41-45 -- and the merge drops them as it does there)
    if ((int)owners[k] != c.rank) matchIndices[k] = ptr::value<Unity<uint2_pair>>(nullptr, (unsigned long)pairCounts[k], gpu);
    else if (matchIndices[k]->getMemoryState() != gpu && matchIndices[k]->getMemoryState() != both) matchIndices[k]->setMemoryState(gpu);
  }
  ncclCheck(ncclGroupStart(), "group");
  for (size_t k = 0; k < owners.size(); ++k)
    if (pairCounts[k])
      ncclCheck(ncclBroadcast(matchIndices[k]->device.get(), matchIndices[k]->device.get(), (size_t)pairCounts[k] * sizeof(uint2_pair), ncclChar,
                              (int)owners[k], c.comm, c.stream), "broadcast of a pair array");
  ncclCheck(ncclGroupEnd(), "group");
  hipCheck(hipStreamSynchronize(c.stream), "pair exchange");
  // ---- replicated merge + KeyPoint table
  return matchFactory.assembleMatchSet(images, features, matchIndices, totalMatches);
}

}  // namespace dist
}  // namespace ssrlcv
