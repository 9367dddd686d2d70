/* oracle/oracle_match.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * CPU restatement of the matching leg of the hot path.
 * Brute force + double-constrained modes and the DMatch / uint2_pair outputs are pinned by the reference's match
 * fixtures (see oracle.h); the F-matrix constrained mode (mode 2) and the Match output kind are PARITY UNPINNED.
 */
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_math.h"

#define EARTH_MAX_KM_FROM_CENT 6384.4  /* include/common_includes.hpp:51 (double) */
#define EARTH_MIN_KM_FROM_CENT 6356.77 /* include/common_includes.hpp:52 (double) */

/* src/Feature.cu:36-42 SIFT_Descriptor::distProtocol (early exit at bestMatch) */
float oracle_dist_protocol(const uint8_t* a, const uint8_t* b, float bestMatch) {
  float dist = 0.0f;
  for (int i = 0; i < 128 && dist < bestMatch; ++i) {
    dist += ((float)a[i] - b[i]) * ((float)a[i] - b[i]);
  }
  return dist;
}

/* src/MatchFactory.cu:1432-1460 getSeedMatchDistances: 32 lanes stride the seed list, lane 0 reduces */
void oracle_seed_distances(uint32_t nq, const o_feature* query, uint32_t ns, const o_feature* seed, float* out) {
#pragma omp parallel for schedule(dynamic, 64)
  for (uint32_t q = 0; q < nq; ++q) {
    float localDist[32];
    for (int l = 0; l < 32; ++l) {
      localDist[l] = FLT_MAX;
      for (uint32_t f = (uint32_t)l; f < ns; f += 32) {
        float d = oracle_dist_protocol(query[q].values, seed[f].values, localDist[l]);
        if (localDist[l] > d) localDist[l] = d;
      }
    }
    float cur = FLT_MAX;
    for (int i = 0; i < 32; ++i)
      if (cur > localDist[i]) cur = localDist[i];
    out[q] = cur;
  }
}

/* src/Image.cu:498-539 getProjectionMatrix + src/matrix_util.cu:34-42 multiply(float3[3], float4[3], float4[3]) */
void oracle_projection_matrix(const o_camera* camera, o_float4 P[3]) {
  o_float3 K[3];
  o_float4 R[3];
  K[0] = f3(camera->foc / camera->dpix.x, 0, camera->size.x / 2.0f);
  K[1] = f3(0, camera->foc / camera->dpix.y, camera->size.y / 2.0f);
  K[2] = f3(0, 0, 1);
  float rx = camera->cam_rot.x, ry = camera->cam_rot.y, rz = camera->cam_rot.z;
  R[0].x = cosf(rz) * cosf(ry);
  R[0].y = sinf(rz) * cosf(ry);
  R[0].z = -1 * sinf(ry);
  R[0].w = 0;
  R[1].x = cosf(rz) * sinf(ry) * sinf(rx) - sinf(rz) * cosf(rx);
  R[1].y = sinf(rz) * sinf(ry) * sinf(rx) + cosf(rz) * cosf(rx);
  R[1].z = cosf(ry) * sinf(rx);
  R[1].w = 0;
  R[2].x = cosf(rz) * sinf(ry) * cosf(rx) + sinf(rz) * sinf(rx);
  R[2].y = sinf(rz) * sinf(ry) * cosf(rx) - cosf(rz) * sinf(rx);
  R[2].z = cosf(ry) * cosf(rx);
  R[2].w = 0;
  o_float3 ecef_cent = f3(camera->cam_pos.x + camera->ecef_offset.x, camera->cam_pos.y + camera->ecef_offset.y,
                          camera->cam_pos.z + camera->ecef_offset.z);
  for (int i = 0; i < 3; i++) {
    R[i].w -= R[i].x * ecef_cent.x + R[i].y * ecef_cent.y + R[i].z * ecef_cent.z;
  }
  for (int r = 0; r < 3; ++r) {
    P[r].x = (K[r].x * R[0].x) + (K[r].y * R[1].x) + (K[r].z * R[2].x);
    P[r].y = (K[r].x * R[0].y) + (K[r].y * R[1].y) + (K[r].z * R[2].y);
    P[r].z = (K[r].x * R[0].z) + (K[r].y * R[1].z) + (K[r].z * R[2].z);
    P[r].w = (K[r].x * R[0].w) + (K[r].y * R[1].w) + (K[r].z * R[2].w);
  }
}

/* src/MatchFactory.cu:1240-1277 getEpipolarEndpoints */
static void epipolar_endpoints(const o_camera* qc, const o_float4* P, o_float2 loc, float delta, o_float2* p1,
                               o_float2* p2) {
  o_float3 queryVec = f3(qc->dpix.x * ((loc.x) - (qc->size.x / 2.0f)), qc->dpix.y * ((loc.y) - (qc->size.y / 2.0f)),
                         qc->foc);
  queryVec = rotate_point(queryVec, qc->cam_rot);
  o_float3 queryCent = f3(qc->cam_pos.x + qc->ecef_offset.x, qc->cam_pos.y + qc->ecef_offset.y,
                          qc->cam_pos.z + qc->ecef_offset.z);
  float a = f3_dot(queryVec, queryVec);
  float b = 2 * f3_dot(queryVec, queryCent);
  /* (EARTH_MAX + delta) is double arithmetic; the subtraction is done in double and narrowed on assignment */
  float c1 = (float)(f3_dot(queryCent, queryCent) -
                     ((EARTH_MAX_KM_FROM_CENT + delta) * (EARTH_MAX_KM_FROM_CENT + delta)));
  float c2 = (float)(f3_dot(queryCent, queryCent) -
                     ((EARTH_MIN_KM_FROM_CENT - delta) * (EARTH_MIN_KM_FROM_CENT - delta)));
  o_float3 tmp = f3_add(f3_lscale((-sqrtf(b * b - 4 * a * c1) - b) / (2 * a), queryVec), queryCent);
  o_float4 X1 = {tmp.x, tmp.y, tmp.z, 1};
  tmp = f3_add(f3_lscale((-sqrtf(b * b - 4 * a * c2) - b) / (2 * a), queryVec), queryCent);
  o_float4 X2 = {tmp.x, tmp.y, tmp.z, 1};
  /* matrix_util.cu:57-61 multiply(float4[3], float4, float3) */
  o_float3 x1 = f3((P[0].x * X1.x) + (P[0].y * X1.y) + (P[0].z * X1.z) + (P[0].w * X1.w),
                   (P[1].x * X1.x) + (P[1].y * X1.y) + (P[1].z * X1.z) + (P[1].w * X1.w),
                   (P[2].x * X1.x) + (P[2].y * X1.y) + (P[2].z * X1.z) + (P[2].w * X1.w));
  o_float3 x2 = f3((P[0].x * X2.x) + (P[0].y * X2.y) + (P[0].z * X2.z) + (P[0].w * X2.w),
                   (P[1].x * X2.x) + (P[1].y * X2.y) + (P[1].z * X2.z) + (P[1].w * X2.w),
                   (P[2].x * X2.x) + (P[2].y * X2.y) + (P[2].z * X2.z) + (P[2].w * X2.w));
  p1->x = x1.x / x1.z;
  p1->y = x1.y / x1.z;
  p2->x = x2.x / x2.z;
  p2->y = x2.y / x2.z;
}

void oracle_epipolar_endpoints(const o_camera* qc, const o_float4* P, const o_float2* loc, float delta, o_float2* p1,
                               o_float2* p2) {
  epipolar_endpoints(qc, P, *loc, delta, p1, p2);
}

/* The scan + lane-0 reduction shared by every matcher kernel
 * (brute force: src/MatchFactory.cu:1462-1506, 2073-2125; double constrained: :2194-2291, :2824-2916; F-matrix
 * constrained, mode 2: :1599-1657, :1710-1775 -- `targetP` then points at the 9 floats of the fundamental matrix). */
void oracle_match_core(int mode, uint32_t nq, const o_feature* query, uint32_t nt, const o_feature* target,
                       const o_camera* queryCam, const o_float4* targetP, float epsilon, float delta,
                       float absoluteThreshold, int* best_index, float* best_dist) {
#pragma omp parallel for schedule(dynamic, 64)
  for (uint32_t q = 0; q < nq; ++q) {
    int localMatch[32];
    float localDist[32];
    o_float2 left = {0, 0}, right = {0, 0}, p1, p2;
    float top = 0, bottom = 0, slope = 0, y_line;
    if (mode == 1) {
      epipolar_endpoints(queryCam, targetP, query[q].loc, delta, &p1, &p2);
      if (p1.x < p2.x) { left = p1; right = p2; } else { left = p2; right = p1; }
      if (left.x == right.x) {
        if (p1.y < p2.y) { top = p1.y; bottom = p2.y; } else { top = p2.y; bottom = p1.y; }
      } else {
        slope = (left.y - right.y) / (left.x - right.x);
      }
    }
    o_float3 epipolar = {0.0f, 0.0f, 0.0f};
    if (mode == 2) {
      const float* fundamental = (const float*)targetP;
      epipolar.x = (fundamental[0] * query[q].loc.x) + (fundamental[1] * query[q].loc.y) + fundamental[2];
      epipolar.y = (fundamental[3] * query[q].loc.x) + (fundamental[4] * query[q].loc.y) + fundamental[5];
      epipolar.z = (fundamental[6] * query[q].loc.x) + (fundamental[7] * query[q].loc.y) + fundamental[8];
    }
    float regEpsilon = epsilon;
    for (int l = 0; l < 32; ++l) {
      localMatch[l] = -1;
      localDist[l] = absoluteThreshold;
      for (uint32_t f = (uint32_t)l; f < nt; f += 32) {
        if (mode == 1) {
          if (target[f].loc.x < left.x - regEpsilon || target[f].loc.x > right.x + regEpsilon) {
            continue;
          } else if (left.x == right.x) {
            if ((top - regEpsilon) > target[f].loc.y || (bottom + regEpsilon) < target[f].loc.y) continue;
          } else {
            y_line = slope * (target[f].loc.x - left.x) + left.y;
            if (fabsf(y_line - target[f].loc.y) > regEpsilon) continue;
          }
        }
        if (mode == 2) {
          /* ax + by + c = 0 */
          float p = -1 * ((epipolar.x * target[f].loc.x) + epipolar.z) / epipolar.y;
          if (fabsf(target[f].loc.y - p) > regEpsilon) continue;
        }
        float d = oracle_dist_protocol(query[q].values, target[f].values, localDist[l]);
        if (localDist[l] > d) {
          localDist[l] = d;
          localMatch[l] = (int)f;
        }
      }
    }
    float cur = absoluteThreshold;
    int matchIndex = -1;
    for (int i = 0; i < 32; ++i) {
      if (cur > localDist[i]) {
        cur = localDist[i];
        matchIndex = localMatch[i];
      }
    }
    best_index[q] = matchIndex;
    best_dist[q] = cur;
  }
}

void oracle_match_dmatch(int mode, uint32_t queryID, uint32_t nq, const o_feature* query, uint32_t targetID, uint32_t nt,
                         const o_feature* target, const o_camera* queryCam, const o_float4* targetP, float epsilon,
                         float delta, const float* seedDistances, float relativeThreshold, float absoluteThreshold,
                         o_dmatch* out) {
  int* idx = (int*)malloc(sizeof(int) * (nq ? nq : 1));
  float* dist = (float*)malloc(sizeof(float) * (nq ? nq : 1));
  oracle_match_core(mode, nq, query, nt, target, queryCam, targetP, epsilon, delta, absoluteThreshold, idx, dist);
  for (uint32_t q = 0; q < nq; ++q) {
    o_dmatch m;
    memset(&m, 0, sizeof m); /* reference leaves keyPoints uninitialised when invalid */
    m.distance = dist[q];
    if (m.distance >= absoluteThreshold || idx[q] == -1) {
      m.invalid = 1;
    } else if (seedDistances && (m.distance / seedDistances[q] > relativeThreshold * relativeThreshold)) {
      m.invalid = 1;
    } else {
      m.invalid = 0;
      m.keyPoints[0].loc = query[q].loc;
      m.keyPoints[1].loc = target[idx[q]].loc;
      m.keyPoints[0].parentId = (int)queryID;
      m.keyPoints[1].parentId = (int)targetID;
    }
    out[q] = m;
  }
  free(idx);
  free(dist);
}

/* Match output (no distance field): matchFeaturesBruteForce :1659-1708 and matchFeaturesConstrained
 * :1710-1775 compare with relativeThreshold, the double-constrained kernel :1777-1873 with its square. */
void oracle_match_match(int mode, uint32_t queryID, uint32_t nq, const o_feature* query, uint32_t targetID, uint32_t nt,
                        const o_feature* target, const o_camera* queryCam, const o_float4* targetP, float epsilon,
                        float delta, const float* seedDistances, float relativeThreshold, float absoluteThreshold,
                        o_match* out) {
  int* idx = (int*)malloc(sizeof(int) * (nq ? nq : 1));
  float* dist = (float*)malloc(sizeof(float) * (nq ? nq : 1));
  oracle_match_core(mode, nq, query, nt, target, queryCam, targetP, epsilon, delta, absoluteThreshold, idx, dist);
  const float limit = mode == 1 ? relativeThreshold * relativeThreshold : relativeThreshold;
  for (uint32_t q = 0; q < nq; ++q) {
    o_match m;
    memset(&m, 0, sizeof m);
    if (dist[q] >= absoluteThreshold || idx[q] == -1) {
      m.invalid = 1;
    } else if (seedDistances && (dist[q] / seedDistances[q] > limit)) {
      m.invalid = 1;
    } else {
      m.invalid = 0;
      m.keyPoints[0].loc = query[q].loc;
      m.keyPoints[1].loc = target[idx[q]].loc;
      m.keyPoints[0].parentId = (int)queryID;
      m.keyPoints[1].parentId = (int)targetID;
    }
    out[q] = m;
  }
  free(idx);
  free(dist);
}

void oracle_match_pairs(int mode, uint32_t queryID, uint32_t nq, const o_feature* query, uint32_t targetID, uint32_t nt,
                        const o_feature* target, const o_camera* queryCam, const o_float4* targetP, float epsilon,
                        float delta, const float* seedDistances, float relativeThreshold, float absoluteThreshold,
                        o_uint2_pair* out) {
  int* idx = (int*)malloc(sizeof(int) * (nq ? nq : 1));
  float* dist = (float*)malloc(sizeof(float) * (nq ? nq : 1));
  oracle_match_core(mode, nq, query, nt, target, queryCam, targetP, epsilon, delta, absoluteThreshold, idx, dist);
  for (uint32_t q = 0; q < nq; ++q) {
    o_uint2_pair m;
    m.a.x = queryID; m.a.y = q; m.b.x = queryID; m.b.y = q; /* a == b marks invalid (MatchFactory.cuh:83-85) */
    if (!(dist[q] >= absoluteThreshold || idx[q] == -1)) {
      /* index-only kernels compare against relativeThreshold, not its square (src/MatchFactory.cu:2907) */
      if (!(seedDistances && (dist[q] / seedDistances[q] > relativeThreshold))) {
        m.b.x = targetID;
        m.b.y = (uint32_t)idx[q];
      }
    }
    out[q] = m;
  }
  free(idx);
  free(dist);
}

/* ---- M6: host-side adjacency merge of generateMatchesExhaustive (src/MatchFactory.cu:943-1020) ----
 * pairs: concatenated validated uint2_pair lists in the reference's pair order (0,1),(0,2)..(1,2)..; pair_counts[p]
 * entries each.  numFeatures[v] per image.  Output: multimatch {n,index} and flattened (image, feature) members.
 * Returns number of multi-matches; *members_out / *mm_out are malloc'd. */
typedef struct { o_uint2* v; uint32_t n, cap; } adj_t;
static void adj_push(adj_t* a, o_uint2 x) {
  if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 2; a->v = (o_uint2*)realloc(a->v, a->cap * sizeof(o_uint2)); }
  a->v[a->n++] = x;
}
static int u2_less(o_uint2 a, o_uint2 b) { /* src/cuda_vec_util.cu:559-563 */
  if (a.x == b.x && a.y == b.y) return 0;
  else if (a.x == b.x) return a.y < b.y;
  else return a.x < b.x;
}
static uint32_t set_intersection_count(const adj_t* A, const adj_t* B) {
  uint32_t i = 0, j = 0, c = 0;
  while (i < A->n && j < B->n) {
    if (u2_less(A->v[i], B->v[j])) ++i;
    else if (u2_less(B->v[j], A->v[i])) ++j;
    else { ++c; ++i; ++j; }
  }
  return c;
}
int oracle_exhaustive_merge(uint32_t numImages, const uint32_t* numFeatures, uint32_t numPairs,
                            const uint32_t* pair_counts, const o_uint2_pair* pairs, o_multimatch** mm_out,
                            o_uint2** members_out, uint32_t* numMembers) {
  adj_t** adj = (adj_t**)calloc(numImages, sizeof(adj_t*));
  for (uint32_t i = 0; i + 1 < numImages; ++i) adj[i] = (adj_t*)calloc(numFeatures[i] ? numFeatures[i] : 1, sizeof(adj_t));
  const o_uint2_pair* p = pairs;
  for (uint32_t k = 0; k < numPairs; ++k)
    for (uint32_t m = 0; m < pair_counts[k]; ++m, ++p) adj_push(&adj[p->a.x][p->a.y], p->b);
  uint32_t cap = 1024, nmm = 0, memcap = 4096, nmem = 0;
  o_multimatch* mm = (o_multimatch*)malloc(cap * sizeof *mm);
  o_uint2* mem = (o_uint2*)malloc(memcap * sizeof *mem);
  for (uint32_t i = 0; i + 1 < numImages; ++i) {
    for (uint32_t f = 0; i + 2 < numImages && f < numFeatures[i]; ++f) {
      adj_t* a = &adj[i][f];
      if (!a->n) continue;
      int bad = 0;
      adj_t* prev = a;
      adj_t* next = NULL;
      for (;;) {
        if (prev->v[0].x == numImages - 1) break;
        next = &adj[prev->v[0].x][prev->v[0].y];
        if (!next->n) break;
        uint32_t inter = set_intersection_count(prev, next);
        if (inter != next->n) { bad = 1; break; }
        else if (next->n == 1) break;
        else prev = next;
      }
      if (bad) { a->n = 0; continue; }
      if (nmm == cap) { cap *= 2; mm = (o_multimatch*)realloc(mm, cap * sizeof *mm); }
      while (nmem + a->n + 1 > memcap) { memcap *= 2; mem = (o_uint2*)realloc(mem, memcap * sizeof *mem); }
      mm[nmm].numKeyPoints = a->n + 1;
      mm[nmm].index = (int)nmem;
      ++nmm;
      mem[nmem].x = i; mem[nmem].y = f; ++nmem;
      for (uint32_t k = 0; k < a->n; ++k) mem[nmem++] = a->v[k];
      for (uint32_t k = 0; k + 1 < a->n; ++k) {
        if (a->v[k].x == numImages - 1) break;
        adj[a->v[k].x][a->v[k].y].n = 0;
      }
    }
  }
  for (uint32_t i = 0; i + 1 < numImages; ++i) {
    for (uint32_t f = 0; f < numFeatures[i]; ++f) free(adj[i][f].v);
    free(adj[i]);
  }
  free(adj);
  *mm_out = mm;
  *members_out = mem;
  *numMembers = nmem;
  return (int)nmm;
}

void oracle_free(void* p) { free(p); }
