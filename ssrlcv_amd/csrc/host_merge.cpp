// ssrlcv_amd/csrc/host_merge.cpp -- host half of MatchFactory<T>::generateMatchesExhaustive
// (src/MatchFactory.cu:943-1020): adjacency lists per (image, feature) in pair order, transitive-consistency walk
// with std::set_intersection, flattening to MultiMatch{n,index} + member list.  Deterministic and single-threaded like
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
  // Adjacency lists in one CSR array instead of upstream's vector<vector<vector<uint2>>> (one heap block per feature:
  // 5 M tiny allocations for four 4096^2 views).  Entries are appended in pair order exactly like upstream's
  // push_back, so every list has the same content and order; `len` is the mutable size (clear() = 0).
  std::vector<size_t> base(V, 0);  // first list slot of image i (images 0..V-2 own lists)
  for (uint32_t i = 0; i + 1 < V; ++i) base[i + 1] = base[i] + numFeatures[i];
  const size_t numLists = base[V - 1];
  std::vector<uint32_t> start(numLists + 1, 0), len(numLists, 0);
  size_t total = 0;
  {
    const ssrlcv_uint2_pair* p = pairs;
    for (uint32_t k = 0; k < numPairs; ++k)
      for (uint32_t m = 0; m < pairCounts[k]; ++m, ++p) {
        // a names a list owner (images 0..V-2), b a feature of a later image: anything else (an un-compacted
        // "invalid" a == b entry of the matcher, an index past a feature array) would index the lists out of bounds
        if (p->a.x >= V - 1 || p->a.y >= numFeatures[p->a.x] || p->b.x >= V || p->b.y >= numFeatures[p->b.x] ||
            p->b.x <= p->a.x)
          return SSRLCV_ERR_INVALID_ARG;
        ++start[base[p->a.x] + p->a.y + 1];
        ++total;
      }
  }
  if (total > 0xffffffffull) return SSRLCV_ERR_CAPACITY;
  for (size_t l = 0; l < numLists; ++l) start[l + 1] += start[l];
  std::vector<U2> entries(total ? total : 1);
  {
    const ssrlcv_uint2_pair* p = pairs;
    for (uint32_t k = 0; k < numPairs; ++k)
      for (uint32_t m = 0; m < pairCounts[k]; ++m, ++p) {
        const size_t l = base[p->a.x] + p->a.y;
        entries[start[l] + len[l]++] = U2{p->b.x, p->b.y};
      }
  }
  auto list_of = [&](uint32_t img, uint32_t feat) { return base[img] + feat; };
  std::vector<ssrlcv_multimatch> mm;
  std::vector<U2> mem;
  mm.reserve(total / 2 + 16);
  mem.reserve(total + total / 2 + 16);
  for (uint32_t i = 0; i + 1 < V; ++i) {
    for (uint32_t f = 0; i + 2 < V && f < numFeatures[i]; ++f) {  // only images 0..V-3 seed multi-matches (:969)
      const size_t a = list_of(i, f);
      if (len[a] == 0) continue;
      bool badMatch = false;
      size_t prev = a;
      while (true) {
        const U2 head = entries[start[prev]];
        if (head.x == V - 1) break;
        const size_t next = list_of(head.x, head.y);
        if (len[next] == 0) break;
        // |set_intersection(list[prev], list[next])| (:984-990) without materialising it: both lists are sorted by the
        // same operator< (an image's list holds at most one entry per later image, appended in pair order), and
        // std::set_intersection's merge is replayed literally
        uint32_t common = 0;
        {
          const U2* p1 = &entries[start[prev]];
          const U2* e1 = p1 + len[prev];
          const U2* p2 = &entries[start[next]];
          const U2* e2 = p2 + len[next];
          while (p1 != e1 && p2 != e2) {
            if (*p1 < *p2) ++p1;
            else if (*p2 < *p1) ++p2;
            else { ++common; ++p1; ++p2; }
          }
        }
        if (common != len[next]) { badMatch = true; break; }
        else if (len[next] == 1) break;
        else prev = next;
      }
      if (badMatch) { len[a] = 0; continue; }
      ssrlcv_multimatch one;
      one.numKeyPoints = len[a] + 1;
      one.index = (int)mem.size();
      mm.push_back(one);
      mem.push_back(U2{i, f});
      mem.insert(mem.end(), entries.begin() + start[a], entries.begin() + start[a] + len[a]);
      for (uint32_t m = 0; m + 1 < len[a]; ++m) {
        const U2 e = entries[start[a] + m];
        if (e.x == V - 1) break;
        len[list_of(e.x, e.y)] = 0;
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
