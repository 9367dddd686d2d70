// ssrlcv_amd/csrc/matcher.hip -- 128-D brute-force descriptor matcher on fp16 MFMA for gfx950
// (SURVEY.md section 8a rows M1-M5).
//
// Formulation.  distProtocol (src/Feature.cu:36-42) is sum((a-b)^2) over 128 u8: an exact integer < 2^24, so
//   dist(q,t) = |q|^2 + ( |t|^2 - 2 q.t )
// is computed exactly by an fp16 x fp16 -> fp32 MFMA when the operands are small integers.  The bracket is produced
// entirely by the matrix core: the target row is stored as -2*t (|values| <= 510, exact in fp16) and the K dimension
// is extended by one 16-wide step that carries |t|^2 as three base-1024 digits against the constants
// {1, 1024, 32768} on the query side (digits < 1024, third digit pre-multiplied by 32: all exact in fp16, every
// partial sum an integer of magnitude < 2^24).  K = 128 + 16 = 144 -> 9 x v_mfma_f32_32x32x16_f16 per 32x32 tile.
//
// Tile orientation.  A operand = 32 targets, B operand = 32 queries, so D[target][query]: a lane owns ONE query
// (col = lane & 31) and its 16 accumulator registers are 16 different targets.  The per-query argmin is then a
// within-lane v_min3 tree (8 VALU ops per 512 pairs) and a wave-uniform branch: only when some lane sees a value that
// could beat (or tie) its running best does the wave take the slow path that decodes indices, applies the epipolar
// prefilter of matchFeaturesDoubleConstrained (src/MatchFactory.cu:2239-2254) and updates a 64-bit key
//   key = dist << 32 | (f mod 32) << 27 | (f / 32)
// whose ordering is exactly the reference's winner rule "smallest distance, then lowest lane f mod 32, then lowest f"
// (32-thread strided scan + lane-0 reduction with strict '>', :2256-2271).  Blocks that split the target range merge
// with a 64-bit atomicMin on that key.
//
// Data layout in the workspace (caller-provided, no hidden allocation):
//   packed queries  [nq_pad][144] fp16   {q_0..q_127, 1, 1024, 32768, 0 x13}
//   packed targets  [nt_pad][144] fp16   {-2 t_0..-2 t_127, n0, n1, 32 n2, 0 x13}; padding rows get a huge norm
//   query norms     [nq_pad] f32, target locations [nt_pad] float2, query epipolar params [nq_pad] x 8 f32, keys [nq_pad] u64
//   modes 1 / 2 only: spatial permutations of both sets (rows above are then in that order), one bounding box per
//   32-target tile and per group of 32 tiles, scratch of the location sort (see "band culling" below)
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include "compact.h"
#include "dev_switch.h"
#include "device_math.h"
#include "spatial_sort.h"
#include "ssrlcv_hip.h"

using namespace sv;

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kKPad = 144;       // fp16 elements per packed row
constexpr int kKSteps = 9;       // 144 / 16
#ifndef SSRLCV_MATCH_QT
#define SSRLCV_MATCH_QT 4
#endif
#ifndef SSRLCV_MATCH_WPS
#define SSRLCV_MATCH_WPS 2
#endif
constexpr int kQT = SSRLCV_MATCH_QT;  // query tiles (of 32) held in registers per wave
constexpr int kWaves = 4;        // waves per block
constexpr int kQPerBlock = kWaves * kQT * 32;  // 512 queries per block
constexpr unsigned long long kNoKey = ~0ull;

struct Geom {  // per-query epipolar segment parameters (mode 1); mode 2 keeps the line (a, b, c) in lo_x, hi_x, left_x
  float lo_x, hi_x;       // left.x - eps, right.x + eps
  float left_x, left_y;
  float slope;
  float top, bottom;      // already widened by eps: top - eps, bottom + eps
  float vertical;         // 1.0 when left.x == right.x
};

__host__ __device__ inline uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }
constexpr uint32_t gcd_u32(uint32_t a, uint32_t b) { return b == 0 ? a : gcd_u32(b, a % b); }
constexpr uint32_t lcm_u32(uint32_t a, uint32_t b) { return a / gcd_u32(a, b) * b; }

// ---- pack ------------------------------------------------------------------------------------------------------
// One wave per 4 features: 16 lanes x 8 bytes per feature.  Writes the fp16 row, the norm and (targets) the location.
// perm (nullable): packed row f holds feature perm[f] (the spatial order of the band-culled modes).
__global__ __launch_bounds__(256) void k_pack(const ssrlcv_sift_feature* __restrict__ feats, uint32_t n, uint32_t n_pad,
                                              int as_target, const uint32_t* __restrict__ perm,
                                              _Float16* __restrict__ packed, float* __restrict__ norms,
                                              ssrlcv_float2* __restrict__ locs) {
  uint32_t f = (blockIdx.x * 256 + threadIdx.x) >> 4;  // 16 lanes per feature
  unsigned sub = threadIdx.x & 15;
  if (f >= n_pad) return;
  _Float16* row = packed + (size_t)f * kKPad;
  if (f < n) {
    const uint32_t src = perm ? perm[f] : f;
    const uint8_t* v = feats[src].values + sub * 8;
    uint2 raw = *reinterpret_cast<const uint2*>(v);  // values[] sits at offset 24 of a 152-byte struct: 8-byte aligned
    uint32_t b[8] = {raw.x & 255u, (raw.x >> 8) & 255u, (raw.x >> 16) & 255u, raw.x >> 24,
                     raw.y & 255u, (raw.y >> 8) & 255u, (raw.y >> 16) & 255u, raw.y >> 24};
    uint32_t nsq = 0;
    half8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      nsq += b[j] * b[j];
      h[j] = as_target ? (_Float16)(-2.0f * (float)b[j]) : (_Float16)(float)b[j];
    }
    *reinterpret_cast<half8*>(row + sub * 8) = h;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) nsq += __shfl_xor(nsq, o, 16);
    if (sub == 0) {
      half8 e0 = {0, 0, 0, 0, 0, 0, 0, 0}, e1 = {0, 0, 0, 0, 0, 0, 0, 0};
      if (as_target) {
        e0[0] = (_Float16)(float)(nsq & 1023u);
        e0[1] = (_Float16)(float)((nsq >> 10) & 1023u);
        e0[2] = (_Float16)(float)((nsq >> 20) * 32u);
        if (locs) locs[f] = feats[src].loc;
      } else {
        e0[0] = (_Float16)1.0f;
        e0[1] = (_Float16)1024.0f;
        e0[2] = (_Float16)32768.0f;
        if (norms) norms[f] = (float)nsq;
      }
      *reinterpret_cast<half8*>(row + 128) = e0;
      *reinterpret_cast<half8*>(row + 136) = e1;
    }
  } else {
    // padding row: zero descriptor; targets get the largest representable norm digits so they never win
    half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    *reinterpret_cast<half8*>(row + sub * 8) = z;
    if (sub == 0) {
      half8 e0 = z;
      if (as_target) {
        e0[2] = (_Float16)60000.0f;  // x 32768 on the query side: ~2e9, above every real distance
        if (locs) { ssrlcv_float2 l; l.x = -1e30f; l.y = -1e30f; locs[f] = l; }
      } else {
        e0[0] = (_Float16)1.0f;
        e0[1] = (_Float16)1024.0f;
        e0[2] = (_Float16)32768.0f;
        if (norms) norms[f] = 0.0f;
      }
      *reinterpret_cast<half8*>(row + 128) = e0;
      *reinterpret_cast<half8*>(row + 136) = z;
    }
  }
}

// ---- epipolar parameters per query (src/MatchFactory.cu:1240-1277 + :2209-2232) ---------------------------------
#define EARTH_MAX_KM_FROM_CENT 6384.4
#define EARTH_MIN_KM_FROM_CENT 6356.77
// geom[s] belongs to query perm[s] (perm nullable = identity); rows nq..nq_pad get a band that meets no box.
__global__ __launch_bounds__(256) void k_geom(const ssrlcv_sift_feature* __restrict__ query, uint32_t nq, uint32_t nq_pad,
                                              const uint32_t* __restrict__ perm, ssrlcv_camera qc, f4 P0, f4 P1, f4 P2,
                                              float epsilon, float delta, Geom* __restrict__ geom) {
  uint32_t q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq_pad) return;
  if (q >= nq) {
    Geom z;
    z.lo_x = FLT_MAX; z.hi_x = -FLT_MAX;
    z.left_x = z.left_y = z.slope = z.top = z.bottom = z.vertical = 0.0f;
    geom[q] = z;
    return;
  }
  ssrlcv_float2 loc = query[perm ? perm[q] : q].loc;
  f3 queryVec = mk3(qc.dpix.x * ((loc.x) - (qc.size.x / 2.0f)), qc.dpix.y * ((loc.y) - (qc.size.y / 2.0f)), qc.foc);
  queryVec = rotate_point(queryVec, qc.cam_rot);
  f3 queryCent = mk3(qc.cam_pos.x + qc.ecef_offset.x, qc.cam_pos.y + qc.ecef_offset.y, qc.cam_pos.z + qc.ecef_offset.z);
  float a = dot(queryVec, queryVec);
  float b = 2 * dot(queryVec, queryCent);
  float c1 = (float)(dot(queryCent, queryCent) - ((EARTH_MAX_KM_FROM_CENT + delta) * (EARTH_MAX_KM_FROM_CENT + delta)));
  float c2 = (float)(dot(queryCent, queryCent) - ((EARTH_MIN_KM_FROM_CENT - delta) * (EARTH_MIN_KM_FROM_CENT - delta)));
  f3 t1 = add(lscale((-sqrtf(b * b - 4 * a * c1) - b) / (2 * a), queryVec), queryCent);
  f3 t2 = add(lscale((-sqrtf(b * b - 4 * a * c2) - b) / (2 * a), queryVec), queryCent);
  float x1x = (P0.x * t1.x) + (P0.y * t1.y) + (P0.z * t1.z) + (P0.w * 1.0f);
  float x1y = (P1.x * t1.x) + (P1.y * t1.y) + (P1.z * t1.z) + (P1.w * 1.0f);
  float x1z = (P2.x * t1.x) + (P2.y * t1.y) + (P2.z * t1.z) + (P2.w * 1.0f);
  float x2x = (P0.x * t2.x) + (P0.y * t2.y) + (P0.z * t2.z) + (P0.w * 1.0f);
  float x2y = (P1.x * t2.x) + (P1.y * t2.y) + (P1.z * t2.z) + (P1.w * 1.0f);
  float x2z = (P2.x * t2.x) + (P2.y * t2.y) + (P2.z * t2.z) + (P2.w * 1.0f);
  float p1x = x1x / x1z, p1y = x1y / x1z, p2x = x2x / x2z, p2y = x2y / x2z;
  float lx, ly, rx, ry;
  if (p1x < p2x) { lx = p1x; ly = p1y; rx = p2x; ry = p2y; } else { lx = p2x; ly = p2y; rx = p1x; ry = p1y; }
  Geom g;
  g.lo_x = lx - epsilon;
  g.hi_x = rx + epsilon;
  g.left_x = lx;
  g.left_y = ly;
  g.slope = 0.0f;
  g.top = 0.0f;
  g.bottom = 0.0f;
  g.vertical = 0.0f;
  if (lx == rx) {
    float top, bottom;
    if (p1y < p2y) { top = p1y; bottom = p2y; } else { top = p2y; bottom = p1y; }
    g.top = top - epsilon;
    g.bottom = bottom + epsilon;
    g.vertical = 1.0f;
  } else {
    g.slope = (ly - ry) / (lx - rx);
  }
  geom[q] = g;
}

// matchFeaturesConstrained's epipolar line of a query (src/MatchFactory.cu:1722-1724)
__global__ __launch_bounds__(256) void k_geom_fundamental(const ssrlcv_sift_feature* __restrict__ query, uint32_t nq,
                                                          uint32_t nq_pad, const uint32_t* __restrict__ perm,
                                                          const float* __restrict__ F9, Geom* __restrict__ geom) {
  uint32_t q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq_pad) return;
  Geom g;
  if (q >= nq) {  // padding: the line y = -3e38 meets no box
    g.lo_x = 0.0f; g.hi_x = 1.0f; g.left_x = 3.0e38f;
    g.left_y = g.slope = g.top = g.bottom = g.vertical = 0.0f;
    geom[q] = g;
    return;
  }
  ssrlcv_float2 loc = query[perm ? perm[q] : q].loc;
  g.lo_x = (F9[0] * loc.x) + (F9[1] * loc.y) + F9[2];
  g.hi_x = (F9[3] * loc.x) + (F9[4] * loc.y) + F9[5];
  g.left_x = (F9[6] * loc.x) + (F9[7] * loc.y) + F9[8];
  g.left_y = g.slope = g.top = g.bottom = g.vertical = 0.0f;
  geom[q] = g;
}

__device__ __forceinline__ bool passes_prefilter(const Geom& g, ssrlcv_float2 t, float epsilon, int mode) {
  if (mode == 2) {  // src/MatchFactory.cu:1735-1737: ax + by + c = 0
    float p = -1 * ((g.lo_x * t.x) + g.left_x) / g.hi_x;
    return !(fabsf(t.y - p) > epsilon);
  }
  // src/MatchFactory.cu:2239-2254
  if (t.x < g.lo_x || t.x > g.hi_x) return false;
  if (g.vertical != 0.0f) {
    if (g.top > t.y || g.bottom < t.y) return false;
  } else {
    float y_line = g.slope * (t.x - g.left_x) + g.left_y;
    if (fabsf(y_line - t.y) > epsilon) return false;
  }
  return true;
}

// ---- band culling (modes 1 / 2) ---------------------------------------------------------------------------------------
// The reference evaluates the prefilter per (query, target) pair and skips the distance when it fails; a brute-force
// MFMA pass computes every distance first.  With both feature sets in spatial order (spatial_sort.hip) a 32-target tile
// has a small bounding box, and a whole tile (or a group of 32 tiles) can be skipped when no query of the wave has a
// band that touches the box.  The test is conservative (margins for the rounding of the per-pair test, NaN geometry
// counts as a hit), the per-pair prefilter still runs on every surviving candidate, so results are unchanged.
struct Box { float x0, y0, x1, y1; };

__global__ __launch_bounds__(256) void k_tile_boxes(const ssrlcv_float2* __restrict__ locT, uint32_t nt, uint32_t numTiles,
                                                    Box* __restrict__ tileBox) {
  uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= numTiles) return;
  Box b = {FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (uint32_t j = 0; j < 32; ++j) {
    uint32_t f = t * 32 + j;
    if (f < nt) {
      ssrlcv_float2 l = locT[f];
      b.x0 = fminf(b.x0, l.x); b.y0 = fminf(b.y0, l.y); b.x1 = fmaxf(b.x1, l.x); b.y1 = fmaxf(b.y1, l.y);
      // a NaN coordinate makes the per-pair test pass (see passes_prefilter): such a tile must never be culled
      if (!(l.x == l.x) || !(l.y == l.y)) { b.x0 = -FLT_MAX; b.y0 = -FLT_MAX; b.x1 = FLT_MAX; b.y1 = FLT_MAX; break; }
    }
  }
  tileBox[t] = b;
}
__global__ __launch_bounds__(256) void k_group_boxes(const Box* __restrict__ tileBox, uint32_t numTiles, uint32_t numGroups,
                                                     Box* __restrict__ groupBox) {
  uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= numGroups) return;
  Box b = {FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (uint32_t j = 0; j < 32; ++j) {
    uint32_t t = g * 32 + j;
    if (t < numTiles) {
      Box tb = tileBox[t];
      b.x0 = fminf(b.x0, tb.x0); b.y0 = fminf(b.y0, tb.y0); b.x1 = fmaxf(b.x1, tb.x1); b.y1 = fmaxf(b.y1, tb.y1);
    }
  }
  groupBox[g] = b;
}

// false only when no target inside `b` can pass passes_prefilter(g, ., epsilon, mode)
__device__ __forceinline__ bool band_hits_box(const Geom& g, const Box& b, float epsilon, int mode) {
  if (mode == 2) {
    const float p0 = -1 * ((g.lo_x * b.x0) + g.left_x) / g.hi_x, p1 = -1 * ((g.lo_x * b.x1) + g.left_x) / g.hi_x;
    const float m = 0.25f + 1e-5f * (fabsf(p0) + fabsf(p1));
    const float lo = fminf(p0, p1) - epsilon - m, hi = fmaxf(p0, p1) + epsilon + m;
    if (!(p0 == p0) || !(p1 == p1)) return true;
    return !(lo > b.y1 || hi < b.y0);
  }
  if (b.x1 < g.lo_x || b.x0 > g.hi_x) return false;  // the per-pair x test, exact; false for NaN bounds
  if (g.vertical != 0.0f) return !(g.top > b.y1 || g.bottom < b.y0);
  const float xa = fmaxf(g.lo_x, b.x0), xb = fminf(g.hi_x, b.x1);
  const float ya = g.slope * (xa - g.left_x) + g.left_y, yb = g.slope * (xb - g.left_x) + g.left_y;
  if (!(ya == ya) || !(yb == yb)) return true;
  const float m = 0.25f + 1e-5f * (fabsf(ya) + fabsf(yb));
  return !(fminf(ya, yb) - epsilon - m > b.y1 || fmaxf(ya, yb) + epsilon + m < b.y0);
}

__device__ __forceinline__ unsigned long long make_key(float dist, uint32_t f) {
  return ((unsigned long long)(uint32_t)dist << 32) | ((unsigned long long)(f & 31u) << 27) | (unsigned long long)(f >> 5);
}

// ---- the contraction ---------------------------------------------------------------------------------------------
// grid.x: query blocks of 512, grid.y: target splits.  256 threads = 4 waves, one per SIMD; each wave keeps kQT query
// tiles (B operands, 36 VGPRs each) resident and streams every target tile of its split through the matrix core.
// BAND: modes 1 / 2 on spatially ordered sets: rows are in perm order, permT maps a packed target row back to the
// caller's index (the key and its tie-break use the caller's index), tiles / groups of tiles are culled by their boxes.
template <bool BAND>
__global__ __launch_bounds__(256, SSRLCV_MATCH_WPS) void k_match(const _Float16* __restrict__ packedQ, const _Float16* __restrict__ packedT,
                                                  const float* __restrict__ normQ, const ssrlcv_float2* __restrict__ locT,
                                                  const Geom* __restrict__ geom, uint32_t nq, uint32_t nt,
                                                  uint32_t tilesPerSplit, int mode, float epsilon, float absThreshold,
                                                  unsigned long long* __restrict__ bestKey,
                                                  const uint32_t* __restrict__ permT, const Box* __restrict__ tileBox,
                                                  const Box* __restrict__ groupBox) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned col = lane & 31, kgrp = lane >> 5;
  const uint32_t qbase = (blockIdx.x * kWaves + wave) * (kQT * 32);
  const uint32_t numTiles = (nt + 31) / 32;
  const uint32_t tile0 = blockIdx.y * tilesPerSplit;
  uint32_t tile1 = tile0 + tilesPerSplit;
  if (tile1 > numTiles) tile1 = numTiles;

  half8 bq[kQT][kKSteps];
  float na[kQT];         // |q|^2 of this lane's query in tile qt
  float bestAcc[kQT];    // fast-path bound in accumulator space (|t|^2 - 2 q.t): anything <= it goes slow
  unsigned long long key[kQT];
#pragma unroll
  for (int qt = 0; qt < kQT; ++qt) {
    uint32_t q = qbase + qt * 32 + col;  // padded rows exist up to nq_pad
    const _Float16* row = packedQ + (size_t)q * kKPad + kgrp * 8;
#pragma unroll
    for (int s = 0; s < kKSteps; ++s) bq[qt][s] = *reinterpret_cast<const half8*>(row + s * 16);
    na[qt] = normQ[q];
    // a candidate needs dist = na + acc < absThreshold; start with a conservative bound (slow path re-checks exactly)
    bestAcc[qt] = (absThreshold >= 3.0e9f) ? FLT_MAX : (ceilf(absThreshold) - na[qt] + 1.0f);
    key[qt] = kNoKey;
  }

  // v_min3_f32 directly: fminf() on MFMA outputs makes hipcc insert a canonicalising v_max per operand
  auto min3 = [](float a, float b, float c) {
    float r;
    asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
  };
  // argmin epilogue of one 32x32 tile (query tile qt): v_min tree, wave-uniform branch, rare slow path
  Geom gq[BAND ? kQT : 1];  // BAND: the wave's query bands stay in registers (culling reads them for every box)
  if (BAND) {
#pragma unroll
    for (int qt = 0; qt < (BAND ? kQT : 1); ++qt) gq[qt] = geom[qbase + qt * 32 + col];
  }
  auto epilogue = [&](uint32_t tt, int qt, const floatx16& acc) {
    float m0 = min3(acc[0], acc[1], acc[2]);
    float m1 = min3(acc[3], acc[4], acc[5]);
    float m2 = min3(acc[6], acc[7], acc[8]);
    float m3 = min3(acc[9], acc[10], acc[11]);
    float m4 = min3(acc[12], acc[13], acc[14]);
    float m = min3(min3(m0, m1, m2), min3(m3, m4, acc[15]), m0);
    if (__any(m <= bestAcc[qt])) {
      // slow path: decode rows.  C/D layout of the 32x32 MFMA: row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
      Geom g;
      if (BAND) g = gq[BAND ? qt : 0];
      else if (mode != 0) g = geom[qbase + qt * 32 + col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[r];
        if (v <= bestAcc[qt]) {
          uint32_t f = tt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kgrp;
          float d = na[qt] + v;  // exact integer
          bool ok = (f < nt) && (d < absThreshold);
          if (ok && mode != 0) ok = passes_prefilter(g, locT[f], epsilon, mode);
          if (ok) {
            unsigned long long k = make_key(d, BAND ? permT[f] : f);
            if (k < key[qt]) {
              key[qt] = k;
              bestAcc[qt] = v;
            }
          }
        }
      }
    }
  };
  // The MFMA chain of query tile qt+1 is issued before the epilogue of tile qt: the matrix pipe runs the 9 MFMAs
  // (9 x 32 cycles) while the VALU does the ~20-instruction min tree of the previous accumulator.
  auto process_tile = [&](uint32_t tt, const half8 (&a)[kKSteps]) {
    floatx16 acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1;
#pragma unroll
    for (int s = 0; s < kKSteps; ++s) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], bq[0][s], acc0, 0, 0, 0);
#pragma unroll
    for (int qt = 1; qt < kQT; ++qt) {
      floatx16& cur = (qt & 1) ? acc1 : acc0;
      floatx16& prev = (qt & 1) ? acc0 : acc1;
      cur = floatx16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < kKSteps; ++s) cur = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], bq[qt][s], cur, 0, 0, 0);
      epilogue(tt, qt - 1, prev);
    }
    // the inline-asm v_min3 of the epilogue is invisible to the back end's MFMA hazard recogniser: give the last chain its
    // wait states by hand (the earlier epilogues run behind the next tile's nine MFMAs)
    asm volatile("s_nop 15");
    asm volatile("s_nop 3");
    epilogue(tt, kQT - 1, ((kQT - 1) & 1) ? acc1 : acc0);
  };
  // Target tiles stream straight from L2 into registers (all blocks walk the same tiles at about the same time, so
  // the 75 MB target set is served by L2 / Infinity Cache).  Staging them through LDS with one barrier per tile
  // measured 25 % slower at one wave per SIMD.
  auto load_tile = [&](uint32_t tt, half8 (&dst)[kKSteps]) {
    const _Float16* trow = packedT + ((size_t)tt * 32 + col) * kKPad + kgrp * 8;
#pragma unroll
    for (int s2 = 0; s2 < kKSteps; ++s2) dst[s2] = *reinterpret_cast<const half8*>(trow + s2 * 16);
  };
  if (BAND) {
    const uint32_t numGroups = (numTiles + 31) / 32;
    for (uint32_t gr = tile0 / 32; gr < numGroups && gr * 32 < tile1; ++gr) {
      const Box gb = groupBox[gr];
      bool gh = false;
#pragma unroll
      for (int qt = 0; qt < (BAND ? kQT : 1); ++qt) gh = gh || band_hits_box(gq[qt], gb, epsilon, mode);
      if (!__any(gh)) continue;
      uint32_t t1 = (gr + 1) * 32;
      if (t1 > tile1) t1 = tile1;
      for (uint32_t tt = gr * 32 > tile0 ? gr * 32 : tile0; tt < t1; ++tt) {
        const Box tb = tileBox[tt];
        unsigned need = 0;
#pragma unroll
        for (int qt = 0; qt < (BAND ? kQT : 1); ++qt)
          if (__any(band_hits_box(gq[qt], tb, epsilon, mode))) need |= 1u << qt;
        if (!need) continue;
        half8 a[kKSteps];
        load_tile(tt, a);
#pragma unroll
        for (int qt = 0; qt < kQT; ++qt) {
          if ((need >> qt) & 1u) {
            floatx16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int s2 = 0; s2 < kKSteps; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s2], bq[qt][s2], acc, 0, 0, 0);
            asm volatile("s_nop 15");  // see process_tile: asm reads of fresh MFMA results
            asm volatile("s_nop 3");
            epilogue(tt, qt, acc);
          }
        }
      }
    }
  } else {
#ifndef SSRLCV_MATCH_DB
#define SSRLCV_MATCH_DB (SSRLCV_MATCH_WPS < 2)
#endif
#if !SSRLCV_MATCH_DB
  // two waves per SIMD: the partner wave hides the load latency, one register buffer suffices
  for (uint32_t tt = tile0; tt < tile1; ++tt) {
    half8 a[kKSteps];
    load_tile(tt, a);
    process_tile(tt, a);
  }
#else
  // one wave per SIMD: two register buffers, the loads of tile tt+1 fly under the 36 MFMAs of tile tt
  half8 bufA[kKSteps], bufB[kKSteps];
  if (tile0 < tile1) load_tile(tile0, bufA);
  for (uint32_t tt = tile0; tt < tile1; tt += 2) {
    if (tt + 1 < tile1) load_tile(tt + 1, bufB);
    process_tile(tt, bufA);
    if (tt + 1 < tile1) {
      if (tt + 2 < tile1) load_tile(tt + 2, bufA);
      process_tile(tt + 1, bufB);
    }
  }
#endif
  }
  // merge: lanes l and l+32 hold the same query (different target rows); other splits merge through the atomic
#pragma unroll
  for (int qt = 0; qt < kQT; ++qt) {
    uint32_t q = qbase + qt * 32 + col;
    if (q < nq && key[qt] != kNoKey) atomicMin(&bestKey[q], key[qt]);
  }
}

#include "matcher_i8.inc"

// ---- finalisation: key -> reference output structs ----------------------------------------------------------------
__global__ __launch_bounds__(256) void k_finalize(const unsigned long long* __restrict__ bestKey, uint32_t nq,
                                                  const ssrlcv_sift_feature* __restrict__ query,
                                                  const ssrlcv_sift_feature* __restrict__ target,
                                                  const float* __restrict__ seedDistances, uint32_t queryID,
                                                  uint32_t targetID, float rel, float absThreshold, int outKind,
                                                  int mode, const uint32_t* __restrict__ permQ, void* __restrict__ out) {
  uint32_t s = blockIdx.x * 256 + threadIdx.x;
  if (s >= nq) return;
  unsigned long long k = bestKey[s];       // keys are in packed-row order
  const uint32_t q = permQ ? permQ[s] : s;  // the caller's query index
  bool found = k != kNoKey;
  float dist = found ? (float)(uint32_t)(k >> 32) : absThreshold;
  uint32_t lo = (uint32_t)(k & 0xffffffffull);
  int matchIndex = found ? (int)(((lo & 0x07ffffffu) << 5) | (lo >> 27)) : -1;
  if (outKind == SSRLCV_OUT_UINT2_PAIR) {
    ssrlcv_uint2_pair m;
    m.a.x = queryID; m.a.y = q; m.b.x = queryID; m.b.y = q;
    if (!(dist >= absThreshold || matchIndex == -1)) {
      bool reject = seedDistances && (dist / seedDistances[q] > rel);  // not squared (src/MatchFactory.cu:2907)
      if (!reject) { m.b.x = targetID; m.b.y = (uint32_t)matchIndex; }
    }
    reinterpret_cast<ssrlcv_uint2_pair*>(out)[q] = m;
    return;
  }
  ssrlcv_dmatch m;
  m.keyPoints[0].parentId = 0; m.keyPoints[0].loc.x = 0; m.keyPoints[0].loc.y = 0;
  m.keyPoints[1] = m.keyPoints[0];
  m.distance = dist;
  if (dist >= absThreshold || matchIndex == -1) {
    m.invalid = 1;
  } else if (seedDistances && (dist / seedDistances[q] > ((outKind == SSRLCV_OUT_MATCH && mode != 1) ? rel : rel * rel))) {
    m.invalid = 1;  // rel^2 except in the brute-force / F-matrix Match kernels (src/MatchFactory.cu:1695, :1762)
  } else {
    m.invalid = 0;
    m.keyPoints[0].loc = query[q].loc;
    m.keyPoints[1].loc = target[matchIndex].loc;
    m.keyPoints[0].parentId = (int)queryID;
    m.keyPoints[1].parentId = (int)targetID;
  }
  if (outKind == SSRLCV_OUT_DMATCH) {
    reinterpret_cast<ssrlcv_dmatch*>(out)[q] = m;
  } else {
    ssrlcv_match mm;
    mm.invalid = m.invalid;
    mm.keyPoints[0] = m.keyPoints[0];
    mm.keyPoints[1] = m.keyPoints[1];
    reinterpret_cast<ssrlcv_match*>(out)[q] = mm;
  }
}

__global__ __launch_bounds__(256) void k_seed_finalize(const unsigned long long* __restrict__ bestKey, uint32_t nq,
                                                       float* __restrict__ out) {
  uint32_t q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  unsigned long long k = bestKey[q];
  out[q] = (k == kNoKey) ? FLT_MAX : (float)(uint32_t)(k >> 32);
}

struct Layout {
  uint32_t nq_pad, nt_pad;
  size_t off_pq, off_pt, off_nq, off_nt, off_lt, off_geom, off_key, off_scratch, off_permq, off_permt, off_tilebox, off_groupbox,
      off_sort, sort_bytes, total;
};

Layout make_layout(uint32_t nq, uint32_t nt) {
  Layout L;
  // a multiple of both kernels' queries per block
  L.nq_pad = round_up(nq ? nq : 1, lcm_u32(kQPerBlock, lcm_u32(kQPerBlock8, kQPerBlock8Band)));
  L.nt_pad = round_up(nt ? nt : 1, 32);
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) / 256 * 256; return r; };
  L.off_pq = take((size_t)L.nq_pad * kKPad * 2);
  L.off_pt = take((size_t)L.nt_pad * kKPad * 2);
  L.off_nq = take((size_t)L.nq_pad * 4);
  L.off_nt = take((size_t)L.nt_pad * 8);  // integer formulation only: |t'|^2 [nt_pad], then -floor(|t'|^2 / 2) [nt_pad]
  L.off_lt = take((size_t)L.nt_pad * 8);
  L.off_geom = take((size_t)L.nq_pad * sizeof(Geom));
  L.off_key = take((size_t)L.nq_pad * 8);
  // scratch for compact_matches: a copy of the largest output struct array + partition counters
  L.off_scratch = take((size_t)nq * sizeof(ssrlcv_dmatch) + svc::workspace_words<1, 8>(nq) * 4 + 256);
  // band-culled modes: spatial permutations, tile / group boxes, sort scratch
  L.off_permq = take((size_t)L.nq_pad * 4);
  L.off_permt = take((size_t)L.nt_pad * 4);
  L.off_tilebox = take((size_t)(L.nt_pad / 32) * sizeof(Box));
  {  // group boxes, and behind them the super-group boxes (32 groups each)
    const size_t groups = (L.nt_pad / 32 + 31) / 32;
    L.off_groupbox = take((groups + (groups + 31) / 32) * sizeof(Box));
  }
  L.sort_bytes = svm::sort_scratch_bytes(nq > nt ? nq : nt);
  L.off_sort = take(L.sort_bytes);
  L.total = o;
  return L;
}

int run_match(const ssrlcv_sift_feature* query, uint32_t nq, const ssrlcv_sift_feature* target, uint32_t nt, int mode,
              const ssrlcv_match_params* p, float absThreshold, char* ws, const Layout& L, hipStream_t st) {
  _Float16* pq = (_Float16*)(ws + L.off_pq);
  _Float16* pt = (_Float16*)(ws + L.off_pt);
  float* nqv = (float*)(ws + L.off_nq);
  ssrlcv_float2* lt = (ssrlcv_float2*)(ws + L.off_lt);
  Geom* geom = (Geom*)(ws + L.off_geom);
  unsigned long long* keys = (unsigned long long*)(ws + L.off_key);
  const bool band = mode != 0;
  uint32_t* permQ = band ? (uint32_t*)(ws + L.off_permq) : nullptr;
  uint32_t* permT = band ? (uint32_t*)(ws + L.off_permt) : nullptr;
  if (band) {
    int rc = svm::sort_by_location(query, nq, permQ, ws + L.off_sort, L.sort_bytes, st);
    if (rc) return rc;
    rc = svm::sort_by_location(target, nt, permT, ws + L.off_sort, L.sort_bytes, st);
    if (rc) return rc;
  }
  // Integer formulation (matcher_i8.inc) by default; SSRLCV_MATCH_F16=1 selects the fp16 one (same results).
  static const bool useF16 = svdev::env("SSRLCV_MATCH_F16") != nullptr;
  if (useF16) {
    hipLaunchKernelGGL(k_pack, dim3((L.nq_pad * 16 + 255) / 256), dim3(256), 0, st, query, nq, L.nq_pad, 0, permQ, pq, nqv,
                       (ssrlcv_float2*)nullptr);
    hipLaunchKernelGGL(k_pack, dim3((L.nt_pad * 16 + 255) / 256), dim3(256), 0, st, target, nt, L.nt_pad, 1, permT, pt,
                       (float*)nullptr, lt);
  } else {
    hipLaunchKernelGGL(k_pack_i8, dim3((L.nq_pad * 16 + 255) / 256), dim3(256), 0, st, query, nq, L.nq_pad, 0, permQ,
                       (uint8_t*)pq, (int*)nqv, (ssrlcv_float2*)nullptr);
    hipLaunchKernelGGL(k_pack_i8, dim3((L.nt_pad * 16 + 255) / 256), dim3(256), 0, st, target, nt, L.nt_pad, 1, permT,
                       (uint8_t*)pt, (int*)(ws + L.off_nt), lt);
  }
  SSRLCV_HIP_TRY(hipMemsetAsync(keys, 0xff, (size_t)L.nq_pad * 8, st));
  float eps = 0.0f;
  if (mode == 1) {
    eps = p->epsilon;
    hipLaunchKernelGGL(k_geom, dim3((L.nq_pad + 255) / 256), dim3(256), 0, st, query, nq, L.nq_pad, permQ, p->queryCamera,
                       p->targetProjection[0], p->targetProjection[1], p->targetProjection[2], p->epsilon, p->delta, geom);
  }
  if (mode == 2) {
    eps = p->epsilon;
    float* F9 = (float*)(ws + L.off_scratch);
    SSRLCV_HIP_TRY(hipMemcpyAsync(F9, p->fundamental, 9 * sizeof(float), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_geom_fundamental, dim3((L.nq_pad + 255) / 256), dim3(256), 0, st, query, nq, L.nq_pad, permQ, F9,
                       geom);
  }
  uint32_t qblocks = L.nq_pad / (useF16 ? kQPerBlock : kQPerBlock8);
  uint32_t numTiles = L.nt_pad / 32;
  const int* ntn = (const int*)(ws + L.off_nt);
  if (band) {
    Box* tileBox = (Box*)(ws + L.off_tilebox);
    Box* groupBox = (Box*)(ws + L.off_groupbox);
    const uint32_t numGroups = (numTiles + 31) / 32;
    hipLaunchKernelGGL(k_tile_boxes, dim3((numTiles + 255) / 256), dim3(256), 0, st, lt, nt, numTiles, tileBox);
    hipLaunchKernelGGL(k_group_boxes, dim3((numGroups + 255) / 256), dim3(256), 0, st, tileBox, numTiles, numGroups,
                       groupBox);
    const uint32_t numSuper = (numGroups + 31) / 32;  // the same union one level up, stored behind the group boxes
    hipLaunchKernelGGL(k_group_boxes, dim3((numSuper + 255) / 256), dim3(256), 0, st, (const Box*)groupBox, numGroups, numSuper,
                       groupBox + numGroups);
    // every wave walks all groups (most are rejected by one box test); target splits only while the grid is small
    const uint32_t qbBand = useF16 ? qblocks : L.nq_pad / kQPerBlock8Band;
    uint32_t splits = 1;
    while (qbBand * splits < 512 && splits * 2 <= numGroups) splits *= 2;
    uint32_t tilesPerSplit = ((numGroups + splits - 1) / splits) * 32;
    if (useF16)
      hipLaunchKernelGGL(k_match<true>, dim3(qblocks, splits), dim3(256), 0, st, pq, pt, nqv, lt, geom, nq, nt,
                         tilesPerSplit, mode, eps, absThreshold, keys, permT, tileBox, groupBox);
    else
      hipLaunchKernelGGL((k_match_i8<true, kQT8Band>), dim3(qbBand, splits), dim3(256), 0, st, (const uint8_t*)pq, (const uint8_t*)pt,
                         (const int*)nqv, ntn, lt, geom, nq, nt, tilesPerSplit, mode, eps, absThreshold, keys, permT,
                         tileBox, groupBox);
  } else {
    // split the target range until the grid has >= 1024 blocks (4 per CU) or tiles run out
    uint32_t splits = 1;
    while (qblocks * splits < 1024 && splits * 2 <= numTiles && numTiles / (splits * 2) >= 16) splits *= 2;
    uint32_t tilesPerSplit = (numTiles + splits - 1) / splits;
    if (useF16)
      hipLaunchKernelGGL(k_match<false>, dim3(qblocks, splits), dim3(256), 0, st, pq, pt, nqv, lt, geom, nq, nt,
                         tilesPerSplit, mode, eps, absThreshold, keys, (const uint32_t*)nullptr, (const Box*)nullptr,
                         (const Box*)nullptr);
    else
      hipLaunchKernelGGL((k_match_i8<false, kQT8Brute>), dim3(qblocks, splits), dim3(256), 0, st, (const uint8_t*)pq, (const uint8_t*)pt,
                         (const int*)nqv, ntn, lt, geom, nq, nt, tilesPerSplit, mode, eps, absThreshold, keys,
                         (const uint32_t*)nullptr, (const Box*)nullptr, (const Box*)nullptr);
  }
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

}  // namespace

namespace {
// compact_matches_async: survivors from the partition's scratch back to the front of the list, count read on the device
__global__ __launch_bounds__(256) void k_copy_counted(uint4* __restrict__ dst, const uint4* __restrict__ src, uint32_t words16PerElem,
                                                      const uint32_t* __restrict__ total, uint32_t* __restrict__ count_out) {
  const size_t n = (size_t)total[0] * words16PerElem;
  if (blockIdx.x == 0 && threadIdx.x == 0) *count_out = total[0];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// generateMatchesExhaustive's KeyPoint table (src/MatchFactory.cu:1007-1020): one thread per member
constexpr int kMaxGatherImages = 64;
struct GatherArgs {
  const ssrlcv_sift_feature* feats[kMaxGatherImages];
  uint32_t count[kMaxGatherImages];
};
// One launch covers the members of kMaxGatherImages consecutive images [imageBase, imageBase + kMaxGatherImages) (their
// feature-array pointers travel in the kernel arguments); more images = more launches over the same member list.
__global__ __launch_bounds__(256) void k_keypoints_from_members(const ssrlcv_uint2* __restrict__ mem, uint32_t n, GatherArgs a,
                                                                uint32_t numImages, uint32_t imageBase, ssrlcv_keypoint* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const ssrlcv_uint2 m = mem[i];
  const uint32_t rel = m.x - imageBase;  // wraps for images below the chunk
  ssrlcv_float2 loc;
  loc.x = loc.y = 0.0f;
  if (m.x < numImages) {
    if (rel >= (uint32_t)kMaxGatherImages) return;  // another launch's member
    if (m.y < a.count[rel]) loc = a.feats[rel][m.y].loc;
  } else if (imageBase != 0) {
    return;  // a member naming no image gets its zero location from the first launch
  }
  // the whole 16-byte element in one store, the 4 padding bytes behind parentId as zeros: a caller that copies the array
  // to the host gets defined bytes without a pass over it
  static_assert(sizeof(ssrlcv_keypoint) == 16, "KeyPoint layout");
  uint4 w;
  w.x = m.x;
  w.y = 0u;
  w.z = __float_as_uint(loc.x);
  w.w = __float_as_uint(loc.y);
  reinterpret_cast<uint4*>(out)[i] = w;
}

// M7: the 2-view MatchSet (src/Pipeline.cu:204-223).  One thread per match; ELEM = ssrlcv_match or ssrlcv_dmatch (the
// end points sit at the same offsets).  Distances are >= 0, so the integer order of their bit patterns is the float order.
template <typename ELEM>
__global__ __launch_bounds__(256) void k_matchset(const ELEM* __restrict__ in, uint32_t n, ssrlcv_keypoint* __restrict__ kp,
                                                  ssrlcv_multimatch* __restrict__ mm, float* __restrict__ maxDistance,
                                                  bool wantMax) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  float d = 0.0f;
  if (i < n) {
    const ELEM m = in[i];
    kp[2 * (size_t)i] = m.keyPoints[0];
    kp[2 * (size_t)i + 1] = m.keyPoints[1];
    ssrlcv_multimatch o;
    o.numKeyPoints = 2u;
    o.index = (int)(2 * i);
    mm[i] = o;
    if (wantMax) d = reinterpret_cast<const ssrlcv_dmatch*>(in)[i].distance;
  }
  if (!wantMax) return;  // uniform
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) d = fmaxf(d, __shfl_xor(d, o, 64));
  if ((threadIdx.x & 63) == 0 && d > 0.0f) atomicMax(reinterpret_cast<int*>(maxDistance), __float_as_int(d));
}

}  // namespace

extern "C" {

size_t ssrlcv_hip_match_workspace_bytes(uint32_t numQuery, uint32_t numTarget) {
  return make_layout(numQuery, numTarget).total;
}

int ssrlcv_hip_seed_distances_u8x128(const ssrlcv_sift_feature* query, uint32_t numQuery, const ssrlcv_sift_feature* seed,
                                     uint32_t numSeed, float* out, void* workspace, size_t workspaceBytes,
                                     ssrlcv_stream_t stream) {
  if (!query || !out || !workspace || (numSeed && !seed)) return SSRLCV_ERR_INVALID_ARG;
  if (numQuery == 0) return SSRLCV_OK;
  Layout L = make_layout(numQuery, numSeed);
  if (workspaceBytes < L.total) return SSRLCV_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int rc = run_match(query, numQuery, seed, numSeed, 0, nullptr, FLT_MAX, (char*)workspace, L, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_seed_finalize, dim3((numQuery + 255) / 256), dim3(256), 0, st,
                     (const unsigned long long*)((char*)workspace + L.off_key), numQuery, out);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

int ssrlcv_hip_match_u8x128(const ssrlcv_sift_feature* query, uint32_t numQuery, const ssrlcv_sift_feature* target,
                            uint32_t numTarget, const float* seedDistances, const ssrlcv_match_params* params_host,
                            int outKind, void* out, void* workspace, size_t workspaceBytes, ssrlcv_stream_t stream) {
  if (!query || !out || !workspace || !params_host || (numTarget && !target)) return SSRLCV_ERR_INVALID_ARG;
  if (outKind < 0 || outKind > 2 || params_host->mode < 0 || params_host->mode > 2) return SSRLCV_ERR_INVALID_ARG;
  if (numQuery == 0) return SSRLCV_OK;
  Layout L = make_layout(numQuery, numTarget);
  if (workspaceBytes < L.total) return SSRLCV_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int rc = run_match(query, numQuery, target, numTarget, params_host->mode, params_host, params_host->absoluteThreshold,
                     (char*)workspace, L, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_finalize, dim3((numQuery + 255) / 256), dim3(256), 0, st,
                     (const unsigned long long*)((char*)workspace + L.off_key), numQuery, query, target, seedDistances,
                     params_host->queryImageID, params_host->targetImageID, params_host->relativeThreshold,
                     params_host->absoluteThreshold, outKind, params_host->mode,
                     params_host->mode != 0 ? (const uint32_t*)((char*)workspace + L.off_permq) : (const uint32_t*)nullptr, out);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

int ssrlcv_hip_compact_matches(int outKind, void* matches, uint32_t numMatches, uint32_t* count_host, void* workspace,
                               size_t workspaceBytes, ssrlcv_stream_t stream) {
  if (!matches || !count_host || !workspace || outKind < 0 || outKind > 2) return SSRLCV_ERR_INVALID_ARG;
  *count_host = 0;
  if (numMatches == 0) return SSRLCV_OK;
  size_t elem = outKind == SSRLCV_OUT_DMATCH ? sizeof(ssrlcv_dmatch)
                : outKind == SSRLCV_OUT_MATCH ? sizeof(ssrlcv_match) : sizeof(ssrlcv_uint2_pair);
  size_t need = (size_t)numMatches * elem + 256 + svc::workspace_words<1, 8>(numMatches) * 4;
  if (workspaceBytes < need) return SSRLCV_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  char* tmp = (char*)workspace;
  uint32_t* words = (uint32_t*)(tmp + ((size_t)numMatches * elem + 255) / 256 * 256);
  uint32_t* totals = nullptr;
  hipError_t e;
  if (outKind == SSRLCV_OUT_UINT2_PAIR) {
    const ssrlcv_uint2_pair* in = (const ssrlcv_uint2_pair*)matches;
    ssrlcv_uint2_pair* o = (ssrlcv_uint2_pair*)tmp;
    auto keyfn = [=] __device__(uint32_t i) -> uint32_t {  // validate (include/MatchFactory.cuh:83-85)
      ssrlcv_uint2_pair m = in[i];
      return (m.a.x == m.b.x && m.a.y == m.b.y) ? 0u : 1u;
    };
    auto emit = [=] __device__(uint32_t i, int, uint32_t d) { o[d] = in[i]; };
    e = svc::partition<1, 8>(numMatches, keyfn, emit, words, &totals, st);
  } else if (outKind == SSRLCV_OUT_DMATCH) {
    const ssrlcv_dmatch* in = (const ssrlcv_dmatch*)matches;
    ssrlcv_dmatch* o = (ssrlcv_dmatch*)tmp;
    auto keyfn = [=] __device__(uint32_t i) -> uint32_t { return in[i].invalid ? 0u : 1u; };
    auto emit = [=] __device__(uint32_t i, int, uint32_t d) { o[d] = in[i]; };
    e = svc::partition<1, 8>(numMatches, keyfn, emit, words, &totals, st);
  } else {
    const ssrlcv_match* in = (const ssrlcv_match*)matches;
    ssrlcv_match* o = (ssrlcv_match*)tmp;
    auto keyfn = [=] __device__(uint32_t i) -> uint32_t { return in[i].invalid ? 0u : 1u; };
    auto emit = [=] __device__(uint32_t i, int, uint32_t d) { o[d] = in[i]; };
    e = svc::partition<1, 8>(numMatches, keyfn, emit, words, &totals, st);
  }
  if (e != hipSuccess) return (int)e;
  uint32_t host_tot[2] = {0, 0};
  SSRLCV_HIP_TRY(hipMemcpyAsync(host_tot, totals, sizeof host_tot, hipMemcpyDeviceToHost, st));
  SSRLCV_HIP_TRY(hipStreamSynchronize(st));
  *count_host = host_tot[0];
  if (host_tot[0]) SSRLCV_HIP_TRY(hipMemcpyAsync(matches, tmp, (size_t)host_tot[0] * elem, hipMemcpyDeviceToDevice, st));
  SSRLCV_HIP_TRY(hipStreamSynchronize(st));
  return SSRLCV_OK;
}

int ssrlcv_hip_compact_matches_async(int outKind, void* matches, uint32_t numMatches, uint32_t* count_dev, void* workspace,
                                     size_t workspaceBytes, ssrlcv_stream_t stream) {
  if (!matches || !count_dev || !workspace || outKind < 0 || outKind > 2) return SSRLCV_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (numMatches == 0) {
    SSRLCV_HIP_TRY(hipMemsetAsync(count_dev, 0, sizeof(uint32_t), st));
    return SSRLCV_OK;
  }
  const size_t elem = outKind == SSRLCV_OUT_DMATCH ? sizeof(ssrlcv_dmatch)
                      : outKind == SSRLCV_OUT_MATCH ? sizeof(ssrlcv_match) : sizeof(ssrlcv_uint2_pair);
  static_assert(sizeof(ssrlcv_dmatch) % 16 == 0 && sizeof(ssrlcv_uint2_pair) % 16 == 0 && sizeof(ssrlcv_match) % 8 == 0, "copy granularity");
  if (elem % 16 != 0 || (reinterpret_cast<size_t>(matches) & 15) != 0) return SSRLCV_ERR_UNSUPPORTED;  // Match (40 B): use the synchronous call
  const size_t need = (size_t)numMatches * elem + 256 + svc::workspace_words<1, 8>(numMatches) * 4;
  if (workspaceBytes < need) return SSRLCV_ERR_WORKSPACE;
  char* tmp = (char*)workspace;
  uint32_t* words = (uint32_t*)(tmp + ((size_t)numMatches * elem + 255) / 256 * 256);
  uint32_t* totals = nullptr;
  hipError_t e;
  if (outKind == SSRLCV_OUT_UINT2_PAIR) {
    const ssrlcv_uint2_pair* in = (const ssrlcv_uint2_pair*)matches;
    ssrlcv_uint2_pair* o = (ssrlcv_uint2_pair*)tmp;
    auto keyfn = [=] __device__(uint32_t i) -> uint32_t {  // validate (include/MatchFactory.cuh:83-85)
      ssrlcv_uint2_pair m = in[i];
      return (m.a.x == m.b.x && m.a.y == m.b.y) ? 0u : 1u;
    };
    auto emit = [=] __device__(uint32_t i, int, uint32_t d) { o[d] = in[i]; };
    e = svc::partition<1, 8>(numMatches, keyfn, emit, words, &totals, st);
  } else {
    const ssrlcv_dmatch* in = (const ssrlcv_dmatch*)matches;
    ssrlcv_dmatch* o = (ssrlcv_dmatch*)tmp;
    auto keyfn = [=] __device__(uint32_t i) -> uint32_t { return in[i].invalid ? 0u : 1u; };
    auto emit = [=] __device__(uint32_t i, int, uint32_t d) { o[d] = in[i]; };
    e = svc::partition<1, 8>(numMatches, keyfn, emit, words, &totals, st);
  }
  if (e != hipSuccess) return (int)e;
  unsigned blocks = (unsigned)(((size_t)numMatches * (elem / 16) + 255) / 256);
  blocks = blocks > 2048u ? 2048u : blocks;
  hipLaunchKernelGGL(k_copy_counted, dim3(blocks), dim3(256), 0, st, (uint4*)matches, (const uint4*)tmp, (uint32_t)(elem / 16), totals,
                     count_dev);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

int ssrlcv_hip_keypoints_from_members(const ssrlcv_uint2* members, uint32_t numMembers,
                                      const ssrlcv_sift_feature* const* features_host, const uint32_t* numFeatures_host,
                                      uint32_t numImages, ssrlcv_keypoint* keyPoints, ssrlcv_stream_t stream) {
  if (!features_host || !numFeatures_host || numImages == 0) return SSRLCV_ERR_INVALID_ARG;
  if (numMembers && (!members || !keyPoints)) return SSRLCV_ERR_INVALID_ARG;
  if (numMembers == 0) return SSRLCV_OK;
  for (uint32_t base = 0; base < numImages; base += (uint32_t)kMaxGatherImages) {
    GatherArgs a;
    for (uint32_t v = 0; v < (uint32_t)kMaxGatherImages; ++v) {
      const uint32_t img = base + v;
      a.feats[v] = img < numImages ? features_host[img] : nullptr;
      a.count[v] = img < numImages && features_host[img] ? numFeatures_host[img] : 0u;
    }
    hipLaunchKernelGGL(k_keypoints_from_members, dim3((numMembers + 255) / 256), dim3(256), 0, (hipStream_t)stream, members,
                       numMembers, a, numImages, base, keyPoints);
  }
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

int ssrlcv_hip_matchset_from_matches(int inKind, const void* matches, uint32_t numMatches, ssrlcv_keypoint* keyPoints,
                                     ssrlcv_multimatch* multiMatches, float* maxDistance, ssrlcv_stream_t stream) {
  if (inKind != SSRLCV_OUT_DMATCH && inKind != SSRLCV_OUT_MATCH) return SSRLCV_ERR_INVALID_ARG;
  if (maxDistance && inKind != SSRLCV_OUT_DMATCH) return SSRLCV_ERR_INVALID_ARG;
  if (numMatches && (!matches || !keyPoints || !multiMatches)) return SSRLCV_ERR_INVALID_ARG;
  if (numMatches > 0x3FFFFFFFu) return SSRLCV_ERR_INVALID_ARG;  // MultiMatch::index is an int
  hipStream_t st = (hipStream_t)stream;
  if (maxDistance) SSRLCV_HIP_TRY(hipMemsetAsync(maxDistance, 0, sizeof(float), st));
  if (numMatches == 0) return SSRLCV_OK;
  const dim3 grid((numMatches + 255) / 256);
  if (inKind == SSRLCV_OUT_DMATCH)
    hipLaunchKernelGGL(k_matchset<ssrlcv_dmatch>, grid, dim3(256), 0, st, (const ssrlcv_dmatch*)matches, numMatches, keyPoints,
                       multiMatches, maxDistance, maxDistance != nullptr);
  else
    hipLaunchKernelGGL(k_matchset<ssrlcv_match>, grid, dim3(256), 0, st, (const ssrlcv_match*)matches, numMatches, keyPoints,
                       multiMatches, maxDistance, false);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

// getProjectionMatrix (src/Image.cu:498-539) + multiply(float3[3], float4[3]) (src/matrix_util.cu:35-42): host arithmetic
void ssrlcv_projection_matrix_host(const ssrlcv_camera* camera, ssrlcv_float4 P[3]) {
  ssrlcv_float3 K[3];
  ssrlcv_float4 R[3];
  K[0].x = camera->foc / camera->dpix.x; K[0].y = 0; K[0].z = camera->size.x / 2.0f;
  K[1].x = 0; K[1].y = camera->foc / camera->dpix.y; K[1].z = camera->size.y / 2.0f;
  K[2].x = 0; K[2].y = 0; K[2].z = 1;
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
  float ex = camera->cam_pos.x + camera->ecef_offset.x, ey = camera->cam_pos.y + camera->ecef_offset.y,
        ez = camera->cam_pos.z + camera->ecef_offset.z;
  for (int i = 0; i < 3; i++) R[i].w -= R[i].x * ex + R[i].y * ey + R[i].z * ez;
  for (int r = 0; r < 3; ++r) {
    P[r].x = (K[r].x * R[0].x) + (K[r].y * R[1].x) + (K[r].z * R[2].x);
    P[r].y = (K[r].x * R[0].y) + (K[r].y * R[1].y) + (K[r].z * R[2].y);
    P[r].z = (K[r].x * R[0].z) + (K[r].y * R[1].z) + (K[r].z * R[2].z);
    P[r].w = (K[r].x * R[0].w) + (K[r].y * R[1].w) + (K[r].z * R[2].w);
  }
}

}  // extern "C"
