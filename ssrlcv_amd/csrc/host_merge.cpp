// ssrlcv_amd/csrc/host_merge.cpp -- host half of MatchFactory<T>::generateMatchesExhaustive
// (src/MatchFactory.cu:943-1020): adjacency lists per (image, feature) in pair order, transitive-consistency walk
// with std::set_intersection, flattening to MultiMatch{n,index} + member list.  Deterministic single-thread STL like
// upstream, so every rank of a multi-GPU run reproduces the same MatchSet from the all-gathered pair arrays.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <vector>
#include "ssrlcv_hip.h"

namespace {
struct U2 {
  uint32_t x, y;
};
inline bool operator==(const U2& a, const U2& b) { return a.x == b.x && a.y == b.y; }
inline bool operator<(const U2& a, const U2& b) {  // src/cuda_vec_util.cu:559-563
  if (a == b) return false;
  else if (a.x == b.x) return a.y < b.y;
  else return a.x < b.x;
}
}  // namespace

extern "C" {

int ssrlcv_merge_matches_host(uint32_t numImages, const uint32_t* numFeatures, uint32_t numPairs, const uint32_t* pairCounts,
                              const ssrlcv_uint2_pair* pairs, ssrlcv_multimatch** matches_out, ssrlcv_uint2** members_out,
                              uint32_t* numMatches, uint32_t* numMembers) {
  if (numImages < 2 || !numFeatures || (numPairs && (!pairCounts || !pairs)) || !matches_out || !members_out || !numMatches ||
      !numMembers)
    return SSRLCV_ERR_INVALID_ARG;
  const uint32_t V = numImages;
  std::vector<std::vector<std::vector<U2>>> adj(V - 1);
  for (uint32_t i = 0; i + 1 < V; ++i) adj[i].resize(numFeatures[i]);
  const ssrlcv_uint2_pair* p = pairs;
  for (uint32_t k = 0; k < numPairs; ++k)
    for (uint32_t m = 0; m < pairCounts[k]; ++m, ++p) {
      if (p->a.x >= V - 1 || p->a.y >= numFeatures[p->a.x]) return SSRLCV_ERR_INVALID_ARG;
      adj[p->a.x][p->a.y].push_back(U2{p->b.x, p->b.y});
    }
  std::vector<ssrlcv_multimatch> mm;
  std::vector<U2> mem;
  for (uint32_t i = 0; i + 1 < V; ++i) {
    for (uint32_t f = 0; i + 2 < V && f < numFeatures[i]; ++f) {  // only images 0..V-3 seed multi-matches (:969)
      std::vector<U2>* a = &adj[i][f];
      if (a->empty()) continue;
      bool badMatch = false;
      std::vector<U2>* prev = a;
      while (true) {
        if (prev->begin()->x == V - 1) break;
        std::vector<U2>* next = &adj[prev->begin()->x][prev->begin()->y];
        if (next->empty()) break;
        std::vector<U2> inter;
        std::set_intersection(prev->begin(), prev->end(), next->begin(), next->end(), std::back_inserter(inter));
        if (inter.size() != next->size()) { badMatch = true; break; }
        else if (next->size() == 1) break;
        else prev = next;
      }
      if (badMatch) { a->clear(); continue; }
      ssrlcv_multimatch one;
      one.numKeyPoints = (uint32_t)a->size() + 1;
      one.index = (int)mem.size();
      mm.push_back(one);
      mem.push_back(U2{i, f});
      mem.insert(mem.end(), a->begin(), a->end());
      for (auto m = a->begin(); m != a->end() - 1; ++m) {
        if (m->x == V - 1) break;
        adj[m->x][m->y].clear();
      }
    }
  }
  *numMatches = (uint32_t)mm.size();
  *numMembers = (uint32_t)mem.size();
  *matches_out = (ssrlcv_multimatch*)std::malloc(sizeof(ssrlcv_multimatch) * (mm.size() ? mm.size() : 1));
  *members_out = (ssrlcv_uint2*)std::malloc(sizeof(ssrlcv_uint2) * (mem.size() ? mem.size() : 1));
  if (!*matches_out || !*members_out) return SSRLCV_ERR_INVALID_ARG;
  if (!mm.empty()) std::memcpy(*matches_out, mm.data(), sizeof(ssrlcv_multimatch) * mm.size());
  if (!mem.empty()) std::memcpy(*members_out, mem.data(), sizeof(ssrlcv_uint2) * mem.size());
  return SSRLCV_OK;
}

void ssrlcv_host_free(void* p) { std::free(p); }

}  // extern "C"
