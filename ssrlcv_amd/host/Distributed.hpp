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
// Round 5: the rest of the flow is sharded here too -- nViewTriangulateSharded (bundle ranges + a grouped broadcast of the
// cloud), selectPairBundles (ssrlcv_hip_select_pair_bundles) and evaluateCameraSetsSharded (the 612-point BA error sweep
// over a bundle range + ncclAllReduce of the sums): the C++ counterpart of ssrlcv_amd/pipeline.py's stages C and BA.
//
// Ordering: every exchange runs on Comm::stream.  The mirror's own calls (SIFT, matcher, ssrlcv_hip_memcpy) are synchronous
// on the null stream and have returned before an exchange is queued, so Comm::stream may be the null stream or any other
// stream: nothing asynchronous is pending when ncclGroupStart is reached.  Within a group RCCL serialises the broadcasts
// on the communicator's stream; "grouped" here means one launch, not concurrency.
//
// Not part of the umbrella header ssrlcv.hpp: a program that includes it links librccl and libamdhip64 (still plain g++:
// rccl_abi.hpp declares the few entry points used); everything else of the mirror needs only the C ABI library.
#pragma once
#include "rccl_abi.hpp"

#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "MatchFactory.hpp"
#include "PointCloudFactory.hpp"

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

// One variable-length exchange: segment k (a device array of `bytes` bytes, allocated on every rank) is filled on every rank
// with rank `owner`'s content.  Two ways to move the bytes, chosen per call by SSRLCV_EXCHANGE (like ssrlcv_amd/dist.py):
//   bcast (default)  one ncclBroadcast per segment at its exact size, all in one group;
//   allgather        ONE ncclAllGather -- the collective the north star names -- of every rank's segments packed back to
//                    back and padded to the largest rank's total (two device-to-device copies per segment on top).
// Same result; which is faster on xGMI is for a world > 1 run to say (bench.py's nview leg reports the mode).
struct Segment { void* ptr; size_t bytes; int owner; };
inline bool exchangeByAllGather() {
  const char* m = std::getenv("SSRLCV_EXCHANGE");
  if (m == nullptr || std::string(m) == "bcast") return false;
  if (std::string(m) == "allgather") return true;
  throw std::invalid_argument("SSRLCV_EXCHANGE must be 'bcast' or 'allgather'");
}
inline void exchangeSegments(Comm& c, const std::vector<Segment>& segs, const char* what) {
  if (c.world == 1) return;
  if (!exchangeByAllGather()) {
    ncclCheck(ncclGroupStart(), "group");
    for (const Segment& s : segs)
      if (s.bytes) ncclCheck(ncclBroadcast(s.ptr, s.ptr, s.bytes, ncclChar, s.owner, c.comm, c.stream), what);
    ncclCheck(ncclGroupEnd(), "group");
    hipCheck(hipStreamSynchronize(c.stream), what);
    return;
  }
  std::vector<size_t> perRank((size_t)c.world, 0);
  for (const Segment& s : segs) perRank[(size_t)s.owner] += s.bytes;
  size_t pad = 16;
  for (size_t b : perRank) pad = std::max(pad, (b + 15) / 16 * 16);
  ptr::device<unsigned char> send((long)pad), recv((long)(pad * (size_t)c.world));
  size_t off = 0;
  for (const Segment& s : segs)
    if (s.owner == c.rank && s.bytes) {
      HipSafeCall(ssrlcv_hip_memcpy(send.get() + off, s.ptr, s.bytes, 2));
      off += s.bytes;
    }
  HipSafeCall(ssrlcv_hip_device_synchronize());  // (the copies ran on the null stream, the collective runs on c.stream)
  ncclCheck(ncclAllGather(send.get(), recv.get(), pad, ncclChar, c.comm, c.stream), what);
  hipCheck(hipStreamSynchronize(c.stream), what);
  std::vector<size_t> cursor((size_t)c.world, 0);
  for (const Segment& s : segs) {
    if (s.owner != c.rank && s.bytes)
      HipSafeCall(ssrlcv_hip_memcpy(s.ptr, recv.get() + (size_t)s.owner * pad + cursor[(size_t)s.owner], s.bytes, 2));
    cursor[(size_t)s.owner] += s.bytes;
  }
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
  {
    std::vector<Segment> segs;
    for (int v = 0; v < V; ++v) segs.push_back({features[(size_t)v]->device.get(), (size_t)nf[(size_t)v] * sizeof(Feature<T>), imageOwner(v, c.world)});
    exchangeSegments(c, segs, "feature exchange");
  }
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
  {
    std::vector<Segment> segs;
    for (size_t k = 0; k < owners.size(); ++k) segs.push_back({matchIndices[k]->device.get(), (size_t)pairCounts[k] * sizeof(uint2_pair), (int)owners[k]});
    exchangeSegments(c, segs, "pair exchange");
  }
  // ---- replicated merge + KeyPoint table
  return matchFactory.assembleMatchSet(images, features, matchIndices, totalMatches);
}

// Stage C of the flow: PointCloudFactory::generateBundles + nViewTriangulate (src/PointCloudFactory.cu:832-925,568-818)
// with the bundles range-partitioned over the ranks -- rank r takes bundles [r * per, (r + 1) * per), per = ceil(M / world),
// the split ssrlcv_amd/dist.py bundle_range makes -- and the cloud completed by one grouped broadcast per rank (exact
// sizes).  matchSet is replicated (every rank holds the same one after generateMatchesExhaustiveSharded); on return every
// rank holds the whole cloud, on the cpu like upstream's triangulators return it, and the MatchSet is back on the cpu
// (generateBundles leaves it there, :909-914).  Bundles are independent, so the cloud is the single-GPU cloud bit for bit.
inline void bundleRange(unsigned long numBundles, int world, int rank, unsigned long& lo, unsigned long& hi) {
  const unsigned long per = (numBundles + (unsigned long)world - 1) / (unsigned long)world;
  lo = std::min((unsigned long)rank * per, numBundles);
  hi = std::min(lo + per, numBundles);
}
inline ptr::value<Unity<float3>> nViewTriangulateSharded(Comm& c, MatchSet* matchSet, std::vector<ptr::value<Image>> images) {
  // (a Unity<T> cannot be empty: a MatchSet without bundles has null members, and there is no cloud to return -- every rank
  // sees the same replicated MatchSet, so all of them leave here together and no collective is left half-entered)
  if (matchSet == nullptr || matchSet->matches == nullptr || matchSet->keyPoints == nullptr || matchSet->matches->size() == 0)
    return ptr::value<Unity<float3>>();
  const unsigned long M = matchSet->matches->size(), K = matchSet->keyPoints->size();
  if (images.empty() || images.at(0)->isPushbroom) throw std::invalid_argument("nViewTriangulateSharded: projective cameras only");
  matchSet->matches->transferMemoryTo(gpu);
  matchSet->keyPoints->transferMemoryTo(gpu);
  unsigned long lo = 0, hi = 0;
  bundleRange(M, c.world, c.rank, lo, hi);
  ptr::value<Unity<float3>> cloud(nullptr, M, gpu);
  if (hi > lo) {
    std::vector<ssrlcv_camera> cams(images.size());
    for (size_t i = 0; i < images.size(); ++i) std::memcpy(&cams[i], &images[i]->camera, sizeof(ssrlcv_camera));
    ptr::device<ssrlcv_camera> cams_d((long)cams.size());
    HipSafeCall(ssrlcv_hip_memcpy(cams_d.get(), cams.data(), cams.size() * sizeof(ssrlcv_camera), 0));
    // MultiMatch::index addresses the shared KeyPoint / line arrays: the range's bundles are a slice of the MultiMatch array
    ptr::device<ssrlcv_bundle> bundles_d((long)(hi - lo));
    ptr::device<ssrlcv_line> lines_d((long)K);
    auto* mm = reinterpret_cast<const ssrlcv_multimatch*>(matchSet->matches->device.get()) + lo;
    auto* kp = reinterpret_cast<const ssrlcv_keypoint*>(matchSet->keyPoints->device.get());
    HipSafeCall(ssrlcv_hip_generate_bundles(mm, kp, (uint32_t)(hi - lo), cams_d.get(), (uint32_t)cams.size(), bundles_d.get(), lines_d.get(),
                                            c.stream));
    const int nview = images.size() > 2 ? 1 : 0;
    auto* pts = reinterpret_cast<ssrlcv_float3*>(cloud->device.get()) + lo;
    if (nview) HipSafeCall(ssrlcv_hip_triangulateN(lines_d.get(), bundles_d.get(), (uint32_t)(hi - lo), pts, nullptr, nullptr, nullptr, 1, c.stream));
    else HipSafeCall(ssrlcv_hip_triangulate2(lines_d.get(), bundles_d.get(), (uint32_t)(hi - lo), pts, nullptr, nullptr, nullptr, c.stream));
    hipCheck(hipStreamSynchronize(c.stream), "triangulation");  // (the device blocks above are released at scope exit)
  }
  if (c.world > 1) {
    std::vector<Segment> segs;
    for (int r = 0; r < c.world; ++r) {
      unsigned long rlo = 0, rhi = 0;
      bundleRange(M, c.world, r, rlo, rhi);
      segs.push_back({cloud->device.get() + rlo, (size_t)(rhi - rlo) * sizeof(float3), r});
    }
    exchangeSegments(c, segs, "cloud exchange");
  }
  matchSet->matches->setFore(gpu);
  matchSet->keyPoints->setFore(gpu);
  matchSet->matches->transferMemoryTo(cpu);
  matchSet->keyPoints->transferMemoryTo(cpu);
  matchSet->matches->clear(gpu);
  matchSet->keyPoints->clear(gpu);
  cloud->transferMemoryTo(cpu);
  cloud->clear(gpu);
  return cloud;
}

// The two-view bundles of image pair (a, b) of an N-view MatchSet as a two-camera MatchSet on the DEVICE
// (ssrlcv_hip_select_pair_bundles): what BundleAdjustTwoView, a two-view method, is given in the N-view flows.
inline MatchSet selectPairBundles(MatchSet* matchSet, int imageA, int imageB) {
  if (matchSet == nullptr || matchSet->matches == nullptr || matchSet->keyPoints == nullptr || matchSet->matches->size() == 0)
    return MatchSet();  // (null members = no bundle of that pair; no zero-length allocation)
  const unsigned long M = matchSet->matches->size(), K = matchSet->keyPoints->size();
  const MemoryState mmOrigin = matchSet->matches->getMemoryState(), kpOrigin = matchSet->keyPoints->getMemoryState();
  if (matchSet->matches->getFore() == cpu) matchSet->matches->transferMemoryTo(gpu);
  if (matchSet->keyPoints->getFore() == cpu) matchSet->keyPoints->transferMemoryTo(gpu);
  ptr::device<ssrlcv_multimatch> mm_d((long)M);
  ptr::device<ssrlcv_keypoint> kp_d((long)(2 * M));
  ptr::device<uint32_t> count_d(1);
  const size_t wsBytes = ssrlcv_hip_select_pair_workspace_bytes((uint32_t)M);
  ptr::device<unsigned char> ws_d((long)wsBytes);
  HipSafeCall(ssrlcv_hip_select_pair_bundles(reinterpret_cast<const ssrlcv_multimatch*>(matchSet->matches->device.get()),
                                             reinterpret_cast<const ssrlcv_keypoint*>(matchSet->keyPoints->device.get()), (uint32_t)M, (uint32_t)K,
                                             imageA, imageB, mm_d.get(), kp_d.get(), count_d.get(), ws_d.get(), wsBytes, nullptr));
  uint32_t n = 0;
  HipSafeCall(ssrlcv_hip_memcpy(&n, count_d.get(), sizeof n, 1));
  if (mmOrigin == cpu) matchSet->matches->clear(gpu);
  if (kpOrigin == cpu) matchSet->keyPoints->clear(gpu);
  MatchSet pair;
  if (n == 0) return pair;  // (a Unity<T> cannot be empty: null members = no bundle of that pair)
  pair.matches = ptr::value<Unity<MultiMatch>>(nullptr, (unsigned long)n, gpu);
  pair.keyPoints = ptr::value<Unity<KeyPoint>>(nullptr, (unsigned long)(2 * n), gpu);
  HipSafeCall(ssrlcv_hip_memcpy(pair.matches->device.get(), mm_d.get(), (size_t)n * sizeof(MultiMatch), 2));
  HipSafeCall(ssrlcv_hip_memcpy(pair.keyPoints->device.get(), kp_d.get(), (size_t)(2 * n) * sizeof(KeyPoint), 2));
  return pair;
}

// The K finite-difference evaluations of one BundleAdjustTwoView iteration (calculateImageGradient + calculateImageHessian,
// src/PointCloudFactory.cu:1059-1504; PointCloudFactory::evaluateCameraSets on one GPU) over this rank's range of the
// pair's bundles, then ncclAllReduce(sum) of the K partial sums.  pairSet: a two-camera MatchSet (selectPairBundles), on
// the device or the host; params_host: K x 12 floats.  The float sum over ranks is not the single-GPU sum to the last bit
// (upstream's own atomicAdd order is not defined either).
inline std::vector<float> evaluateCameraSetsSharded(Comm& c, MatchSet* pairSet, std::vector<ptr::value<Image>> twoImages,
                                                    const std::vector<float>& params_host, uint32_t K) {
  std::vector<float> sums((size_t)K, 0.0f);
  ptr::device<float> sums_d((long)K);
  HipSafeCall(ssrlcv_hip_memset(sums_d.get(), 0, (size_t)K * sizeof(float)));
  // the memset runs on the null stream and is not documented as host-synchronous: c.stream may be a non-blocking stream,
  // and a rank without bundles queues nothing else in front of the all-reduce -- order the two explicitly
  HipSafeCall(ssrlcv_hip_device_synchronize());
  if (pairSet->matches != nullptr && pairSet->matches->size()) {
    if (pairSet->matches->getFore() == cpu) pairSet->matches->transferMemoryTo(gpu);
    if (pairSet->keyPoints->getFore() == cpu) pairSet->keyPoints->transferMemoryTo(gpu);
    unsigned long lo = 0, hi = 0;
    bundleRange(pairSet->matches->size(), c.world, c.rank, lo, hi);
    if (hi > lo) {
      std::vector<ssrlcv_camera> cams(twoImages.size());
      for (size_t i = 0; i < twoImages.size(); ++i) std::memcpy(&cams[i], &twoImages[i]->camera, sizeof(ssrlcv_camera));
      ptr::device<ssrlcv_camera> cams_d((long)cams.size());
      HipSafeCall(ssrlcv_hip_memcpy(cams_d.get(), cams.data(), cams.size() * sizeof(ssrlcv_camera), 0));
      ptr::device<float> params_d((long)params_host.size());
      HipSafeCall(ssrlcv_hip_memcpy(params_d.get(), params_host.data(), params_host.size() * sizeof(float), 0));
      HipSafeCall(ssrlcv_hip_ba_sweep2(reinterpret_cast<const ssrlcv_multimatch*>(pairSet->matches->device.get()) + lo,
                                       reinterpret_cast<const ssrlcv_keypoint*>(pairSet->keyPoints->device.get()), (uint32_t)(hi - lo), cams_d.get(),
                                       (uint32_t)cams.size(), params_d.get(), K, sums_d.get(), nullptr, 0, c.stream));
      hipCheck(hipStreamSynchronize(c.stream), "BA sweep");
    }
  }
  if (c.world > 1) {
    ncclCheck(ncclAllReduce(sums_d.get(), sums_d.get(), (size_t)K, ncclFloat32, ncclSum, c.comm, c.stream), "all-reduce of the BA sums");
    hipCheck(hipStreamSynchronize(c.stream), "BA all-reduce");
  }
  HipSafeCall(ssrlcv_hip_memcpy(sums.data(), sums_d.get(), (size_t)K * sizeof(float), 1));
  return sums;
}

}  // namespace dist
}  // namespace ssrlcv
