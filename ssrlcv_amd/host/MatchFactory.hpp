// ssrlcv_amd/host/MatchFactory.hpp -- MatchFactory<T> with the reference's signatures (include/MatchFactory.cuh:23-309),
// bound to the fp16-MFMA matcher of the HIP C ABI.  Provided for T = SIFT_Descriptor (the pipeline's only
// instantiation, src/Pipeline.cu:175); Window_* descriptors, FeatureMatch outputs, disparity matchers and match-file
// IO are out of scope.
//
// Every method keeps upstream's memory-state contract: inputs are moved to the gpu for the call and restored to their
// origin state before returning; results are fresh Unity objects on the gpu holding only the valid matches
// (validateMatches = stable compaction).
#pragma once
#include <algorithm>
#include <iterator>
#include <vector>
#include "Feature.hpp"
#include "Image.hpp"

namespace ssrlcv {

struct uint2_pair { uint2 a; uint2 b; };
struct KeyPoint { int parentId; float2 loc; };
struct MultiMatch { unsigned int numKeyPoints; int index; };
struct MatchSet {
  ptr::value<Unity<KeyPoint>> keyPoints;
  ptr::value<Unity<MultiMatch>> matches;
};
struct Match { bool invalid; KeyPoint keyPoints[2]; };
struct DMatch : Match { float distance; };

static_assert(sizeof(uint2_pair) == sizeof(ssrlcv_uint2_pair) && sizeof(KeyPoint) == sizeof(ssrlcv_keypoint) &&
              sizeof(MultiMatch) == sizeof(ssrlcv_multimatch) && sizeof(Match) == sizeof(ssrlcv_match) &&
              sizeof(DMatch) == sizeof(ssrlcv_dmatch), "match POD layouts");

template <typename T>
class MatchFactory {
 private:
  ptr::value<Unity<Feature<T>>> seedFeatures;
  // grow-only matcher workspace kept for the factory's lifetime: generateMatchesExhaustive calls run() once per image
  // pair and getSeedDistances once per image, and a hipMalloc / hipFree of ~100 MB per call was a fifth of a pair's time
  ptr::device<unsigned char> wsCache;
  size_t wsCacheBytes = 0;
  unsigned char* workspace(size_t bytes) {
    if (bytes > wsCacheBytes) {
      wsCache = ptr::device<unsigned char>((long)bytes);
      wsCacheBytes = bytes;
    }
    return wsCache.get();
  }

  static ssrlcv_match_params make_params(int mode, ptr::value<Image> query, ptr::value<Image> target, float epsilon,
                                         float delta, float rel, float absolute) {
    ssrlcv_match_params p;
    std::memset(&p, 0, sizeof p);
    p.mode = mode;
    p.queryImageID = (uint32_t)query->id;
    p.targetImageID = (uint32_t)target->id;
    p.epsilon = epsilon;
    p.delta = delta;
    p.relativeThreshold = rel;
    p.absoluteThreshold = absolute;
    std::memcpy(&p.queryCamera, &query->camera, sizeof(ssrlcv_camera));
    if (mode == 1) {
      ssrlcv_camera tc;
      std::memcpy(&tc, &target->camera, sizeof tc);
      ssrlcv_projection_matrix_host(&tc, p.targetProjection);  // getProjectionMatrix (src/Image.cu:498-539)
    }
    return p;
  }

  // common body of generate*Matches*: launch, validate (compact), shrink
  template <typename OUT>
  ptr::value<Unity<OUT>> run(int mode, int outKind, ptr::value<Image> query, ptr::value<Unity<Feature<T>>> queryFeatures,
                             ptr::value<Image> target, ptr::value<Unity<Feature<T>>> targetFeatures, float epsilon,
                             float delta, ptr::value<Unity<float>> seedDistances, const float* fundamental = nullptr) {
    static_assert(std::is_same<T, SIFT_Descriptor>::value, "MatchFactory is provided for SIFT_Descriptor");
    MemoryState origin[2] = {queryFeatures->getMemoryState(), targetFeatures->getMemoryState()};
    if (origin[0] != gpu) queryFeatures->setMemoryState(gpu);
    if (origin[1] != gpu) targetFeatures->setMemoryState(gpu);
    uint32_t nq = (uint32_t)queryFeatures->size(), nt = (uint32_t)targetFeatures->size();
    const float* seed_d = nullptr;
    MemoryState seedOrigin = null;
    if (seedDistances != nullptr) {
      if (seedDistances->size() != queryFeatures->size()) {
        logger.err << "ERROR: seedDistances should have come from matching a seed image to queryFeatures";
        std::exit(-1);
      }
      seedOrigin = seedDistances->getMemoryState();
      if (seedOrigin != gpu) seedDistances->setMemoryState(gpu);
      seed_d = seedDistances->device.get();
    }
    ssrlcv_match_params p = make_params(mode, query, target, epsilon, delta, relativeThreshold, absoluteThreshold);
    if (fundamental) std::memcpy(p.fundamental, fundamental, sizeof p.fundamental);
    size_t wsBytes = ssrlcv_hip_match_workspace_bytes(nq, nt);
    unsigned char* ws = workspace(wsBytes);
    ptr::value<Unity<OUT>> matches(nullptr, (unsigned long)nq, gpu);
    HipSafeCall(ssrlcv_hip_match_u8x128(reinterpret_cast<const ssrlcv_sift_feature*>(queryFeatures->device.get()), nq,
                                        reinterpret_cast<const ssrlcv_sift_feature*>(targetFeatures->device.get()), nt,
                                        seed_d, &p, outKind, matches->device.get(), ws, wsBytes, nullptr));
    HipCheckError();
    if (seedDistances != nullptr && seedOrigin != gpu) seedDistances->setMemoryState(seedOrigin);
    // validateMatches (src/MatchFactory.cu:32-108)
    uint32_t left = 0;
    HipSafeCall(ssrlcv_hip_compact_matches(outKind, matches->device.get(), nq, &left, ws, wsBytes, nullptr));
    if (left == 0) {
      logger.info << "No valid matches found";
    } else {
      logger.info.printf("%d valid matches found out of %lu original matches", (int)left, (unsigned long)nq);
      ptr::device<OUT> validated((long)left);
      HipSafeCall(ssrlcv_hip_memcpy(validated.get(), matches->device.get(), (size_t)left * sizeof(OUT), 2));
      matches->setData(validated, left, gpu);
    }
    if (origin[0] != gpu) queryFeatures->setMemoryState(origin[0]);
    if (origin[1] != gpu) targetFeatures->setMemoryState(origin[1]);
    return matches;
  }

 public:
  float absoluteThreshold;
  float relativeThreshold;

  MatchFactory(float relativeThreshold, float absoluteThreshold)
      : absoluteThreshold(absoluteThreshold), relativeThreshold(relativeThreshold) {
    seedFeatures = nullptr;
  }
  void setSeedFeatures(ptr::value<Unity<Feature<T>>> seedFeatures) { this->seedFeatures = seedFeatures; }
  bool hasSeedFeatures() const { return seedFeatures != nullptr; }

  // src/MatchFactory.cu:315-346
  ptr::value<Unity<float>> getSeedDistances(ptr::value<Unity<Feature<T>>> features) {
    MemoryState origin = features->getMemoryState();
    if (seedFeatures->getMemoryState() != gpu) seedFeatures->setMemoryState(gpu);
    if (origin != gpu) features->setMemoryState(gpu);
    uint32_t nq = (uint32_t)features->size(), ns = (uint32_t)seedFeatures->size();
    ptr::value<Unity<float>> out(nullptr, (unsigned long)nq, gpu);
    size_t wsBytes = ssrlcv_hip_match_workspace_bytes(nq, ns);
    HipSafeCall(ssrlcv_hip_seed_distances_u8x128(reinterpret_cast<const ssrlcv_sift_feature*>(features->device.get()), nq,
                                                 reinterpret_cast<const ssrlcv_sift_feature*>(seedFeatures->device.get()),
                                                 ns, out->device.get(), workspace(wsBytes), wsBytes, nullptr));
    HipCheckError();
    if (origin != gpu) features->setMemoryState(origin);
    return out;
  }

  // brute force (src/MatchFactory.cu:504-547, :754-800)
  ptr::value<Unity<DMatch>> generateDistanceMatches(ptr::value<Image> query, ptr::value<Unity<Feature<T>>> queryFeatures,
                                                    ptr::value<Image> target, ptr::value<Unity<Feature<T>>> targetFeatures,
                                                    ptr::value<Unity<float>> seedDistances = nullptr) {
    return run<DMatch>(0, SSRLCV_OUT_DMATCH, query, queryFeatures, target, targetFeatures, 0.0f, 0.0f, seedDistances);
  }
  ptr::value<Unity<uint2_pair>> generateMatchesIndexOnly(ptr::value<Image> query, ptr::value<Unity<Feature<T>>> queryFeatures,
                                                         ptr::value<Image> target, ptr::value<Unity<Feature<T>>> targetFeatures,
                                                         ptr::value<Unity<float>> seedDistances = nullptr) {
    return run<uint2_pair>(0, SSRLCV_OUT_UINT2_PAIR, query, queryFeatures, target, targetFeatures, 0.0f, 0.0f, seedDistances);
  }
  // orbit double-constrained (src/MatchFactory.cu:599-651, :852-905)
  ptr::value<Unity<DMatch>> generateDistanceMatchesDoubleConstrained(ptr::value<Image> query, ptr::value<Unity<Feature<T>>> queryFeatures,
                                                                     ptr::value<Image> target, ptr::value<Unity<Feature<T>>> targetFeatures,
                                                                     float epsilon, float delta,
                                                                     ptr::value<Unity<float>> seedDistances = nullptr) {
    return run<DMatch>(1, SSRLCV_OUT_DMATCH, query, queryFeatures, target, targetFeatures, epsilon, delta, seedDistances);
  }
  ptr::value<Unity<uint2_pair>> generateMatchesDoubleConstrainedIndexOnly(ptr::value<Image> query, ptr::value<Unity<Feature<T>>> queryFeatures,
                                                                          ptr::value<Image> target, ptr::value<Unity<Feature<T>>> targetFeatures,
                                                                          float epsilon, float delta,
                                                                          ptr::value<Unity<float>> seedDistances = nullptr) {
    return run<uint2_pair>(1, SSRLCV_OUT_UINT2_PAIR, query, queryFeatures, target, targetFeatures, epsilon, delta, seedDistances);
  }

  // Match outputs (src/MatchFactory.cu:349-395 brute force, :447-503 double-constrained): the pose-estimation stage's
  // matcher (src/Pipeline.cu:94)
  ptr::value<Unity<Match>> generateMatches(ptr::value<Image> query, ptr::value<Unity<Feature<T>>> queryFeatures,
                                           ptr::value<Image> target, ptr::value<Unity<Feature<T>>> targetFeatures,
                                           ptr::value<Unity<float>> seedDistances = nullptr) {
    return run<Match>(0, SSRLCV_OUT_MATCH, query, queryFeatures, target, targetFeatures, 0.0f, 0.0f, seedDistances);
  }
  ptr::value<Unity<Match>> generateMatchesDoubleConstrained(ptr::value<Image> query, ptr::value<Unity<Feature<T>>> queryFeatures,
                                                            ptr::value<Image> target, ptr::value<Unity<Feature<T>>> targetFeatures,
                                                            float epsilon, float delta,
                                                            ptr::value<Unity<float>> seedDistances = nullptr) {
    return run<Match>(1, SSRLCV_OUT_MATCH, query, queryFeatures, target, targetFeatures, epsilon, delta, seedDistances);
  }

  // F-matrix constrained (src/MatchFactory.cu:397-445, :549-597, :802-850): candidates within epsilon px of the
  // epipolar line F (x, y, 1) of the query feature
  ptr::value<Unity<Match>> generateMatchesConstrained(ptr::value<Image> query, ptr::value<Unity<Feature<T>>> queryFeatures,
                                                      ptr::value<Image> target, ptr::value<Unity<Feature<T>>> targetFeatures,
                                                      float epsilon, float fundamental[3][3],
                                                      ptr::value<Unity<float>> seedDistances = nullptr) {
    return run<Match>(2, SSRLCV_OUT_MATCH, query, queryFeatures, target, targetFeatures, epsilon, 0.0f, seedDistances, &fundamental[0][0]);
  }
  ptr::value<Unity<DMatch>> generateDistanceMatchesConstrained(ptr::value<Image> query, ptr::value<Unity<Feature<T>>> queryFeatures,
                                                               ptr::value<Image> target, ptr::value<Unity<Feature<T>>> targetFeatures,
                                                               float epsilon, float fundamental[3][3],
                                                               ptr::value<Unity<float>> seedDistances = nullptr) {
    return run<DMatch>(2, SSRLCV_OUT_DMATCH, query, queryFeatures, target, targetFeatures, epsilon, 0.0f, seedDistances, &fundamental[0][0]);
  }
  ptr::value<Unity<uint2_pair>> generateMatchesConstrainedIndexOnly(ptr::value<Image> query, ptr::value<Unity<Feature<T>>> queryFeatures,
                                                                    ptr::value<Image> target, ptr::value<Unity<Feature<T>>> targetFeatures,
                                                                    float epsilon, float fundamental[3][3],
                                                                    ptr::value<Unity<float>> seedDistances = nullptr) {
    return run<uint2_pair>(2, SSRLCV_OUT_UINT2_PAIR, query, queryFeatures, target, targetFeatures, epsilon, 0.0f, seedDistances, &fundamental[0][0]);
  }

  // convertMatchToRaw (src/MatchFactory.cu:257-280, :2921-2926): slice the DMatch base
  ptr::value<Unity<Match>> getRawMatches(ptr::value<Unity<DMatch>> matches) {
    MemoryState origin = matches->getMemoryState();
    bool onGpu = origin == gpu || matches->getFore() == gpu;
    if (onGpu && origin == gpu) matches->transferMemoryTo(cpu);
    ptr::host<Match> raw((long)matches->size());
    for (unsigned long i = 0; i < matches->size(); ++i) raw.get()[i] = Match(matches->host.get()[i]);
    if (onGpu && origin == gpu) matches->clear(cpu);
    ptr::value<Unity<Match>> out(raw, matches->size(), cpu);
    if (onGpu) out->setMemoryState(gpu);
    return out;
  }

  // src/MatchFactory.cu:907-1028: all pairs i<j on the GPU, adjacency merge on the host
  MatchSet generateMatchesExhaustive(std::vector<ptr::value<Image>> images, std::vector<ptr::value<Unity<Feature<T>>>> features,
                                     float epsilon, float delta, bool ordered = true, float estimatedOverlap = 0.0f) {
    MatchSet matchSet;
    matchSet.keyPoints = nullptr;
    matchSet.matches = nullptr;
    if (estimatedOverlap == 0) logger.warn << "WARNING: estimated overlap fraction of 0.0f requires unordered match interpolation";
    std::vector<ptr::value<Unity<uint2_pair>>> matchIndices;
    ptr::value<Unity<float>> seedDistances;
    unsigned long long totalMatches = 0;
    logger.info << "matching images";
    int skipCounter = 0;
    for (size_t q = 0; q + 1 < images.size(); ++q) {
      if (seedFeatures != nullptr) seedDistances = getSeedDistances(features[q]);
      for (size_t t = q + 1; t < images.size(); ++t) {
        if (ordered && estimatedOverlap > 0.0f && ++skipCounter * (1 - estimatedOverlap) > 1.0f) continue;
        matchIndices.push_back(generateMatchesDoubleConstrainedIndexOnly(images[q], features[q], images[t], features[t],
                                                                         epsilon, delta, seedDistances));  // GEO_ORBIT == 1
        totalMatches += matchIndices.back()->size();
      }
    }
    return assembleMatchSet(images, features, matchIndices, totalMatches);
  }

  // The host half of generateMatchesExhaustive (src/MatchFactory.cu:938-1028) from the validated uint2_pair arrays of
  // every image pair in upstream's pair order: the merge and the KeyPoint table.  Public because the sharded driver
  // (host/Distributed.hpp) arrives here with pair arrays it matched on several GPUs and exchanged.
  MatchSet assembleMatchSet(std::vector<ptr::value<Image>>& images, std::vector<ptr::value<Unity<Feature<T>>>>& features,
                            std::vector<ptr::value<Unity<uint2_pair>>>& matchIndices, unsigned long long totalMatches) {
    MatchSet matchSet;
    matchSet.keyPoints = nullptr;
    matchSet.matches = nullptr;
    if (totalMatches == 0) {
      logger.err << "There were no matches found in the set of images, likely due to unreasonable threshold";
      logger.err << "exiting...";
      std::exit(0);
    }
    // the merge (adjacency lists + consistency walk, :943-1005) and the KeyPoint table (:1007-1020) live in the library,
    // shared with the multi-GPU driver: on the device for up to 32 images (ssrlcv_hip_merge_matches: the same MatchSet as
    // upstream's single-threaded walk, the match arrays never leave the GPU), on the host beyond (ssrlcv_merge_matches_host)
    const size_t V = images.size();
    std::vector<uint32_t> numFeatures(V), pairCounts;
    for (size_t i = 0; i < V; ++i) numFeatures[i] = (uint32_t)features[i]->size();
    for (auto& m : matchIndices) pairCounts.push_back((uint32_t)m->size());
    const size_t mergeBytes = ssrlcv_hip_merge_workspace_bytes((uint32_t)V, numFeatures.data(), (uint32_t)totalMatches);
    if (mergeBytes != 0 && totalMatches <= 0x7fffffffull) {
      std::vector<MemoryState> origin(V);
      std::vector<const ssrlcv_sift_feature*> featPtr(V);
      for (size_t i = 0; i < V; ++i) {
        origin[i] = features[i]->getMemoryState();
        if (origin[i] != gpu && origin[i] != both) features[i]->setMemoryState(gpu);
        else if (origin[i] == both && features[i]->getFore() == cpu) features[i]->transferMemoryTo(gpu);
        featPtr[i] = reinterpret_cast<const ssrlcv_sift_feature*>(features[i]->device.get());
      }
      ptr::device<uint2_pair> allPairs((long)totalMatches);
      size_t at = 0;
      for (auto& m : matchIndices) {
        if (m->size() == 0) continue;
        if (m->getMemoryState() != gpu && m->getMemoryState() != both) m->setMemoryState(gpu);
        HipSafeCall(ssrlcv_hip_memcpy(allPairs.get() + at, m->device.get(), m->size() * sizeof(uint2_pair), 2));
        at += m->size();
      }
      ptr::device<unsigned char> mergeWs((long)mergeBytes);
      ptr::device<MultiMatch> mm_d((long)totalMatches);
      ptr::device<uint2> mem_d((long)(2 * totalMatches));
      ptr::device<uint32_t> counts_d(4);
      HipSafeCall(ssrlcv_hip_merge_matches((uint32_t)V, numFeatures.data(), (uint32_t)pairCounts.size(), pairCounts.data(),
                                           reinterpret_cast<const ssrlcv_uint2_pair*>(allPairs.get()), mergeWs.get(), mergeBytes,
                                           reinterpret_cast<ssrlcv_multimatch*>(mm_d.get()),
                                           reinterpret_cast<ssrlcv_uint2*>(mem_d.get()), counts_d.get(), nullptr));
      uint32_t counts[4] = {0, 0, 0, 0};  // {matches, members, input status, rounds}; the copy synchronises the (asynchronous) merge
      HipSafeCall(ssrlcv_hip_memcpy(counts, counts_d.get(), sizeof counts, 1));
      if (counts[2]) HipSafeCall(SSRLCV_ERR_INVALID_ARG);  // a malformed pair list (cannot come out of the matcher)
      const uint32_t nmm = counts[0], nmem = counts[1];
      logger.info.printf("total matches found in set = %d", (int)nmm);
      matchSet.matches = ptr::value<Unity<MultiMatch>>(nullptr, (unsigned long)nmm, cpu);
      if (nmm) HipSafeCall(ssrlcv_hip_memcpy(matchSet.matches->host.get(), mm_d.get(), sizeof(MultiMatch) * nmm, 1));
      matchSet.keyPoints = ptr::value<Unity<KeyPoint>>(nullptr, (unsigned long)nmem, gpu);
      if (nmem)
        HipSafeCall(ssrlcv_hip_keypoints_from_members(reinterpret_cast<const ssrlcv_uint2*>(mem_d.get()), nmem, featPtr.data(),
                                                      numFeatures.data(), (uint32_t)V,
                                                      reinterpret_cast<ssrlcv_keypoint*>(matchSet.keyPoints->device.get()), nullptr));
      HipCheckError();
      for (size_t i = 0; i < V; ++i)
        if (origin[i] != gpu && origin[i] != both) features[i]->setMemoryState(origin[i]);
      return matchSet;
    }
    std::vector<uint2_pair> allPairs;
    for (auto& m : matchIndices) {
      if (m->getMemoryState() != cpu) m->setMemoryState(cpu);
      allPairs.insert(allPairs.end(), m->host.get(), m->host.get() + m->size());
    }
    ssrlcv_multimatch* mm_raw = nullptr;
    ssrlcv_uint2* mem_raw = nullptr;
    uint32_t nmm = 0, nmem = 0;
    HipSafeCall(ssrlcv_merge_matches_host((uint32_t)V, numFeatures.data(), (uint32_t)pairCounts.size(), pairCounts.data(),
                                          reinterpret_cast<const ssrlcv_uint2_pair*>(allPairs.data()), &mm_raw, &mem_raw,
                                          &nmm, &nmem));
    std::vector<MemoryState> origin(V);
    for (size_t i = 0; i < V; ++i) {
      origin[i] = features[i]->getMemoryState();
      if (origin[i] != cpu && origin[i] != both) features[i]->setMemoryState(cpu);
      else if (origin[i] == both && features[i]->getFore() == gpu) features[i]->transferMemoryTo(cpu);
    }
    logger.info.printf("total matches found in set = %d", (int)nmm);
    matchSet.matches = ptr::value<Unity<MultiMatch>>(nullptr, (unsigned long)nmm, cpu);
    std::memcpy(matchSet.matches->host.get(), mm_raw, sizeof(MultiMatch) * nmm);
    std::vector<KeyPoint> kp_vec(nmem);
    for (uint32_t k = 0; k < nmem; ++k)
      kp_vec[k] = {(int)mem_raw[k].x, features[mem_raw[k].x]->host.get()[mem_raw[k].y].loc};
    ssrlcv_host_free(mm_raw);
    ssrlcv_host_free(mem_raw);
    matchSet.keyPoints = ptr::value<Unity<KeyPoint>>(nullptr, (unsigned long)kp_vec.size(), gpu);
    HipSafeCall(ssrlcv_hip_memcpy(matchSet.keyPoints->device.get(), kp_vec.data(), kp_vec.size() * sizeof(KeyPoint), 0));
    for (size_t i = 0; i < V; ++i)
      if (origin[i] != cpu && origin[i] != both) features[i]->setMemoryState(origin[i]);
    return matchSet;
  }
};

}  // namespace ssrlcv
