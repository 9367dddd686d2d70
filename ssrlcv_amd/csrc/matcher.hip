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
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
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
#ifndef SSRLCV_MATCH_F16_ORDER
#define SSRLCV_MATCH_F16_ORDER 1
#endif
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
// MFMA pass computes every distance first.  Here a whole 32-target tile (a group of 32 tiles, a super-group of 32
// groups) is skipped when no query of the wave has a band that can reach it.  The test is conservative (margins for
// every rounding involved, non-finite geometry counts as a hit) and the per-pair prefilter still runs on every
// surviving candidate, so results are unchanged; what the order of the two sets decides is only how much is culled.
//
// Round 4: the frame of the pair.  Both sets used to be ordered by (64-pixel image-row strip, x).  A wave's 32 queries
// then sit in a 20 x 64 pixel patch of the QUERY image, their bands -- near-parallel segments of the target image --
// spread over up to 64 pixels across the band direction, and a target tile is 64 pixels across as well: the wave computed
// every target within ~90 pixels of its bands where a query accepts 25 (epsilon), three of four candidates failed the
// per-pair test in the epilogue's slow path.  Now
//   * u = the dominant direction of the queries' bands in the target image (axial mean, k_band_direction); the frame
//     (w, v) = (u . p, u_perp . p) has w along the bands and v across them;
//   * targets are ordered by (v strip, w), strips 0.4 epsilon wide within [4, 16] px: a tile is one strip across the bands
//     and long along them;
//   * queries are ordered by WHERE THEIR BAND LIES in that frame (v strip of the band's centre, then its w), not by where
//     they are in their own image: the 32 bands of a wave nearly coincide;
//   * boxes and bands are compared in the frame: a band is the line v = vc + slope (w - wc), |w - wc| <= hw, thickened by
//     `half` (BandR, formed in double precision from the Geom of the exact test), a box is centre + half extents, and the
//     test is 8 vector instructions with no per-test margin arithmetic (the margins are folded into hw / half / the boxes).
// With horizontal epipolar lines (the bench scenes) this equals image-row strips; at 16 pixels those alone took the match stage of
// the 4 x 4096^2 flow from 22.4 to 14.6 ms; the frame makes that independent of the direction of the baseline.
struct Frame {        // device-resident, written by k_band_direction / k_target_keys
  float ux, uy;       // unit vector along the bands (target image)
  int wminI, wmaxI, vminI, vmaxI;  // order-preserving integer images of the targets' frame bounds (ord_of)
};
struct BandR { float wc, hw, slope, vc, half, pad0, pad1, pad2; };  // 32 bytes
struct Box { float wc, vc, hw, hv; };                                // centre and half extents in the frame

__device__ __forceinline__ int ord_of(float f) {  // monotone float -> int
  int i = __builtin_bit_cast(int, f);
  return i >= 0 ? i : (int)(0x80000000u - (unsigned)i);
}
__device__ __forceinline__ float of_ord(int i) {
  return __builtin_bit_cast(float, i >= 0 ? i : (int)(0x80000000u - (unsigned)i));
}
__device__ __forceinline__ bool finite_f(float x) { return fabsf(x) <= FLT_MAX; }

#ifndef SSRLCV_BAND_STRIP
#define SSRLCV_BAND_STRIP 16.0f
#endif

// u from the axial mean of the band directions: sum of (cos 2a, sin 2a) over a sample of the queries.  One block; the
// summation order is fixed, so u is reproducible.  forceDeg < 1e9 (developer builds, tests): take that direction instead.
__global__ __launch_bounds__(1024) void k_band_direction(const Geom* __restrict__ geomU, uint32_t nq, int mode, float forceDeg,
                                                         Frame* __restrict__ frame) {
  __shared__ double s_c[1024], s_s[1024];
  double c2 = 0.0, s2 = 0.0;
  const uint32_t stride = (nq + 4095u) / 4096u;  // at most 4096 samples: four per thread (one block: latency, not work)
  for (uint32_t i = threadIdx.x * (stride ? stride : 1); i < nq; i += 1024u * (stride ? stride : 1)) {
    const Geom g = geomU[i];
    double dx, dy;  // a vector along the band
    if (mode == 2) { dx = g.hi_x; dy = -(double)g.lo_x; }
    else if (g.vertical != 0.0f) { dx = 0.0; dy = 1.0; }
    else { dx = 1.0; dy = g.slope; }
    const double n2 = dx * dx + dy * dy;
    if (n2 > 0.0 && n2 < 1e300) {  // (false for NaN / inf)
      c2 += (dx * dx - dy * dy) / n2;
      s2 += 2.0 * dx * dy / n2;
    }
  }
  s_c[threadIdx.x] = c2;
  s_s[threadIdx.x] = s2;
  __syncthreads();
  for (unsigned o = 512; o > 0; o >>= 1) {
    if (threadIdx.x < o) { s_c[threadIdx.x] += s_c[threadIdx.x + o]; s_s[threadIdx.x] += s_s[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double a = 0.5 * atan2(s_s[0], s_c[0]);  // atan2(0, 0) = 0: no dominant direction -> the image rows
    if (forceDeg < 1e9f) a = (double)forceDeg * (3.14159265358979323846 / 180.0);
    float ux = (float)cos(a), uy = (float)sin(a);
    if (!finite_f(ux) || !finite_f(uy)) { ux = 1.0f; uy = 0.0f; }
    frame->ux = ux;
    frame->uy = uy;
    frame->wminI = frame->vminI = 0x7fffffff;
    frame->wmaxI = frame->vmaxI = (int)0x80000000;
  }
}

// (strip << 16) | position: both clamped to 16 bits around an offset of 32 768 (frame coordinates of any image up to
// 23 000 pixels on a side fit; beyond, keys saturate -- any order is correct, only culling suffers).  NaN -> 0.
__device__ __forceinline__ uint32_t frame_key(float w, float v, float invStrip) {
  const float vs = fminf(fmaxf(floorf(v * invStrip) + 32768.0f, 0.0f), 65535.0f);
  const float wp = fminf(fmaxf(floorf(w) + 32768.0f, 0.0f), 65535.0f);
  return ((uint32_t)vs << 16) | (uint32_t)wp;
}

// sort keys of the targets; mode 2 (infinite bands) also needs the bounds of the target set in the frame: per-block
// reduction + 4 atomics per block (same-address atomics serialise at ~50 ns: one set per WAVE cost 1.3 ms for 413 000 targets)
__global__ __launch_bounds__(256) void k_target_keys(const ssrlcv_sift_feature* __restrict__ feats, uint32_t n, Frame* __restrict__ frame,
                                                     float invStrip, int wantBounds, uint32_t* __restrict__ keys) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const float ux = frame->ux, uy = frame->uy;
  int w0 = 0x7fffffff, w1 = (int)0x80000000, v0 = 0x7fffffff, v1 = (int)0x80000000;
  if (i < n) {
    const ssrlcv_float2 l = feats[i].loc;
    const float w = ux * l.x + uy * l.y, v = ux * l.y - uy * l.x;
    keys[i] = frame_key(w, v, invStrip);
    if (finite_f(w) && finite_f(v)) { w0 = w1 = ord_of(w); v0 = v1 = ord_of(v); }
  }
  if (!wantBounds) return;  // (uniform)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    w0 = min(w0, __shfl_xor(w0, o, 64)); w1 = max(w1, __shfl_xor(w1, o, 64));
    v0 = min(v0, __shfl_xor(v0, o, 64)); v1 = max(v1, __shfl_xor(v1, o, 64));
  }
  __shared__ int s_b[4][4];
  if ((threadIdx.x & 63) == 0) {
    const int wv = threadIdx.x >> 6;
    s_b[wv][0] = w0; s_b[wv][1] = w1; s_b[wv][2] = v0; s_b[wv][3] = v1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) {
      w0 = min(w0, s_b[k][0]); w1 = max(w1, s_b[k][1]); v0 = min(v0, s_b[k][2]); v1 = max(v1, s_b[k][3]);
    }
    if (w0 <= w1) {
      atomicMin(&frame->wminI, w0); atomicMax(&frame->wmaxI, w1);
      atomicMin(&frame->vminI, v0); atomicMax(&frame->vmaxI, v1);
    }
  }
}

// The band of one query in the frame, from the Geom its exact per-pair test uses.  Real-number derivation (the frame map
// is LINEAR in the float values ux, uy taken as exact): a point that passes the test is (x, L(x) + e) with
// lo_x <= x <= hi_x, |e| <= epsilon, L(x) = slope (x - left_x) + left_y.  With A = ux + uy slope, B = ux slope - uy:
//   w = A x + Cw + uy e,  v = B x + Cv + ux e   =>   v = s' (w - wc) + vc + e (ux - s' uy),  s' = B / A,
// so the band is the image of the centre segment, `half` = epsilon |ux - s' uy| thick in v, and
// |w - wc| <= |A| (hi_x - lo_x) / 2 + |uy| epsilon.  A band nearly across the frame (|A| small: an outlier of the pair's
// geometry) and the reference's `vertical` case (a rectangle) are bounded by their frame-aligned hull instead (slope 0).
// Margins: 0.25 px absolute (the exact test's own float rounding at image-sized coordinates), 1e-5 of every magnitude
// that enters the test (huge coordinates), and the conversion to float.  Non-finite input -> a band that meets
// everything (the per-pair test passes on NaN); lo_x > hi_x (padding rows) -> a band that meets nothing.
__device__ BandR make_bandr(const Geom& g, int mode, float epsilon, float uxf, float uyf, float targetRadius) {
  BandR r;
  r.pad0 = r.pad1 = r.pad2 = 0.0f;
  const float kInf = __builtin_inff();
  auto meets_all = [&]() { r.wc = 0.0f; r.hw = kInf; r.slope = 0.0f; r.vc = 0.0f; r.half = kInf; };
  const double ux = uxf, uy = uyf, e = fabs((double)epsilon);
  if (!(fabsf(epsilon) <= FLT_MAX)) { meets_all(); return r; }
  if (mode == 2) {  // a x + b y + c = 0 as (lo_x, hi_x, left_x): |y + (a x + c) / b| <= epsilon, all x
    const double a = g.lo_x, b = g.hi_x, c = g.left_x;
    if (!(fabs(a) <= 1e300) || !(fabs(b) <= 1e300) || !(fabs(c) <= 1e300) || b == 0.0) { meets_all(); return r; }
    const double s = -a / b, y0 = -c / b;
    const double A = ux + uy * s, B = ux * s - uy;
    if (!(fabs(A) >= 0.05 * sqrt(1.0 + s * s))) { meets_all(); return r; }  // an infinite band across the frame
    const double sp = B / A;
    const double vc = ux * y0 - sp * (uy * y0);  // the image of (0, y0), moved along the line to w = 0
    const double gm = 1e-5 * (fabs(y0) * (1.0 + fabs(s)) + fabs(vc)) + 1e-5 * (fabs(s) + fabs(sp)) * (double)targetRadius;
    r.wc = 0.0f; r.hw = kInf; r.slope = (float)sp; r.vc = (float)vc;
    r.half = (float)(e * fabs(ux - sp * uy) + 0.25 + gm);
    if (!finite_f(r.slope) || !finite_f(r.vc) || !finite_f(r.half)) meets_all();
    return r;
  }
  const double lo = g.lo_x, hi = g.hi_x;
  if (!(fabs(lo) <= 1e300) || !(fabs(hi) <= 1e300)) { meets_all(); return r; }
  if (lo > hi) { r.wc = 0.0f; r.hw = -kInf; r.slope = 0.0f; r.vc = 0.0f; r.half = 0.0f; return r; }
  double w0, v0, w1, v1, sp = 0.0, half, mag;
  bool hull = false;
  if (g.vertical != 0.0f) {  // the rectangle [lo, hi] x [top, bottom] (both already widened by epsilon)
    const double t = g.top, bt = g.bottom;
    if (!(fabs(t) <= 1e300) || !(fabs(bt) <= 1e300)) { meets_all(); return r; }
    const double cw[4] = {ux * lo + uy * t, ux * lo + uy * bt, ux * hi + uy * t, ux * hi + uy * bt};
    const double cv[4] = {ux * t - uy * lo, ux * bt - uy * lo, ux * t - uy * hi, ux * bt - uy * hi};
    w0 = fmin(fmin(cw[0], cw[1]), fmin(cw[2], cw[3])); w1 = fmax(fmax(cw[0], cw[1]), fmax(cw[2], cw[3]));
    v0 = fmin(fmin(cv[0], cv[1]), fmin(cv[2], cv[3])); v1 = fmax(fmax(cv[0], cv[1]), fmax(cv[2], cv[3]));
    mag = fabs(lo) + fabs(hi) + fabs(t) + fabs(bt);
    const double gm = 1e-5 * mag;
    r.wc = (float)(0.5 * (w0 + w1)); r.hw = (float)(0.5 * (w1 - w0) + 1e-3 + gm);
    r.slope = 0.0f;
    r.vc = (float)(0.5 * (v0 + v1)); r.half = (float)(0.5 * (v1 - v0) + 0.25 + gm);
    if (!finite_f(r.wc) || !finite_f(r.vc)) meets_all();
    return r;
  }
  const double s = g.slope, xl = g.left_x, yl = g.left_y;
  if (!(fabs(s) <= 1e300) || !(fabs(xl) <= 1e300) || !(fabs(yl) <= 1e300)) { meets_all(); return r; }
  const double ylo = s * (lo - xl) + yl, yhi = s * (hi - xl) + yl;
  w0 = ux * lo + uy * ylo; v0 = ux * ylo - uy * lo;
  w1 = ux * hi + uy * yhi; v1 = ux * yhi - uy * hi;
  const double A = ux + uy * s, B = ux * s - uy;
  mag = fabs(lo) + fabs(hi) + fabs(xl) + fabs(yl) + fabs(ylo) + fabs(yhi);
  const double gm = 1e-5 * mag;
  if (fabs(A) >= 0.05 * sqrt(1.0 + s * s)) {
    sp = B / A;
    half = e * fabs(ux - sp * uy) + 0.25 + gm;
  } else {
    hull = true;
    half = 0.5 * fabs(v1 - v0) + e * fabs(ux) + 0.25 + gm;
  }
  r.wc = (float)(0.5 * (w0 + w1));
  r.hw = (float)(0.5 * fabs(w1 - w0) + e * fabs(uy) + 1e-3 + gm);
  r.slope = hull ? 0.0f : (float)sp;
  r.vc = (float)(0.5 * (v0 + v1));
  r.half = (float)half;
  if (!finite_f(r.wc) || !finite_f(r.vc) || !finite_f(r.slope) || !(r.half == r.half) || !(r.hw == r.hw)) meets_all();
  return r;
}

__device__ __forceinline__ float target_radius(const Frame* frame) {  // max |w|, |v| over the finite targets (0 if none)
  const int a = frame->wminI, b = frame->wmaxI, c = frame->vminI, d = frame->vmaxI;
  if (a > b) return 0.0f;
  return fmaxf(fmaxf(fabsf(of_ord(a)), fabsf(of_ord(b))), fmaxf(fabsf(of_ord(c)), fabsf(of_ord(d))));
}

// sort keys of the queries: the strip of the band's centre across the frame, then its position along it.  An infinite
// band (mode 2) is placed by its v at the middle of the target set.
__global__ __launch_bounds__(256) void k_query_keys(const Geom* __restrict__ geomU, uint32_t nq, int mode, float epsilon,
                                                    const Frame* __restrict__ frame, float invStrip,
                                                    uint32_t* __restrict__ keys) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nq) return;
  const BandR r = make_bandr(geomU[i], mode, epsilon, frame->ux, frame->uy, target_radius(frame));
  float w = r.wc, v = r.vc;
  if (mode == 2 && frame->wminI <= frame->wmaxI) {
    const float wm = 0.5f * (of_ord(frame->wminI) + of_ord(frame->wmaxI));
    v = fmaf(r.slope, wm - r.wc, r.vc);
    w = r.slope * 4096.0f;  // near-equal lines side by side
  }
  keys[i] = frame_key(w, v, invStrip);
}

// bands of the packed query rows (geom is in packed order, rows nq .. nq_pad hold k_geom's padding)
__global__ __launch_bounds__(256) void k_bandr(const Geom* __restrict__ geom, uint32_t nq, uint32_t nq_pad, int mode, float epsilon,
                                               const Frame* __restrict__ frame, BandR* __restrict__ bandr) {
  const uint32_t q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq_pad) return;
  BandR r;
  if (q < nq) {
    r = make_bandr(geom[q], mode, epsilon, frame->ux, frame->uy, target_radius(frame));
  } else {
    r.wc = 0.0f; r.hw = -__builtin_inff(); r.slope = 0.0f; r.vc = 0.0f; r.half = 0.0f;
    r.pad0 = r.pad1 = r.pad2 = 0.0f;
  }
  bandr[q] = r;
}

// min / max form while a box is being formed; centre form in memory.  The centre form is inflated by the rounding of the
// frame map (1e-6 of |x| + |y|) and of its own arithmetic, so that |w - wc| <= hw, |v - vc| <= hv hold for the real-number
// images of the targets.
struct BoxMM { float w0, v0, w1, v1, mag; bool all; };
__device__ __forceinline__ Box to_box(const BoxMM& m) {
  Box b;
  if (m.all) { b.wc = 0.0f; b.vc = 0.0f; b.hw = FLT_MAX; b.hv = FLT_MAX; return b; }
  if (m.w0 > m.w1) { b.wc = 0.0f; b.vc = 0.0f; b.hw = -FLT_MAX; b.hv = -FLT_MAX; return b; }  // empty: meets nothing finite
  const float slack = 1e-3f + 2e-6f * m.mag;
  b.wc = 0.5f * m.w0 + 0.5f * m.w1; b.hw = fminf((0.5f * m.w1 - 0.5f * m.w0) + slack, FLT_MAX);
  b.vc = 0.5f * m.v0 + 0.5f * m.v1; b.hv = fminf((0.5f * m.v1 - 0.5f * m.v0) + slack, FLT_MAX);
  return b;
}

// tilePar (nullable, integer formulation): bit j = parity of |t'|^2 of the tile's row j (the slow path's exact distance)
__global__ __launch_bounds__(256) void k_tile_boxes(const ssrlcv_float2* __restrict__ locT, uint32_t nt, uint32_t numTiles,
                                                    const Frame* __restrict__ frame, Box* __restrict__ tileBox,
                                                    const int* __restrict__ normT, uint32_t* __restrict__ tilePar) {
  uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= numTiles) return;
  const float ux = frame->ux, uy = frame->uy;
  BoxMM m = {FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, 0.0f, false};
  for (uint32_t j = 0; j < 32; ++j) {
    uint32_t f = t * 32 + j;
    if (f < nt) {
      const ssrlcv_float2 l = locT[f];
      const float w = ux * l.x + uy * l.y, v = ux * l.y - uy * l.x;
      // a NaN coordinate makes the per-pair test pass (see passes_prefilter): such a tile must never be culled
      if (!finite_f(l.x) || !finite_f(l.y) || !finite_f(w) || !finite_f(v)) { m.all = true; break; }
      m.w0 = fminf(m.w0, w); m.w1 = fmaxf(m.w1, w); m.v0 = fminf(m.v0, v); m.v1 = fmaxf(m.v1, v);
      m.mag = fmaxf(m.mag, fabsf(l.x) + fabsf(l.y));
    }
  }
  tileBox[t] = to_box(m);
  if (tilePar) {
    uint32_t par = 0;
    for (uint32_t j = 0; j < 32; ++j) par |= ((uint32_t)normT[t * 32 + j] & 1u) << j;  // (rows exist up to nt_pad)
    tilePar[t] = par;
  }
}
__global__ __launch_bounds__(256) void k_group_boxes(const Box* __restrict__ tileBox, uint32_t numTiles, uint32_t numGroups,
                                                     Box* __restrict__ groupBox) {
  uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= numGroups) return;
  BoxMM m = {FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, 0.0f, false};
  for (uint32_t j = 0; j < 32; ++j) {
    uint32_t t = g * 32 + j;
    if (t < numTiles) {
      const Box tb = tileBox[t];
      if (tb.hw < 0.0f) continue;  // empty
      if (tb.hw >= FLT_MAX || tb.hv >= FLT_MAX) { m.all = true; break; }
      m.w0 = fminf(m.w0, tb.wc - tb.hw); m.w1 = fmaxf(m.w1, tb.wc + tb.hw);
      m.v0 = fminf(m.v0, tb.vc - tb.hv); m.v1 = fmaxf(m.v1, tb.vc + tb.hv);
      m.mag = fmaxf(m.mag, fabsf(tb.wc) + tb.hw + fabsf(tb.vc) + tb.hv);
    }
  }
  groupBox[g] = to_box(m);
}

// false only when no target inside `b` can pass passes_prefilter for the query of band `g` (see make_bandr).  NaN -> true.
__device__ __forceinline__ bool band_hits_box(const BandR& g, const Box& b) {
  const float d = b.wc - g.wc;
  if (fabsf(d) > b.hw + g.hw) return false;
  const float dv = fmaf(g.slope, d, g.vc) - b.vc;
  return !(fabsf(dv) > fmaf(fabsf(g.slope), b.hw, g.half + b.hv));
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
                                                  const Box* __restrict__ groupBox, const BandR* __restrict__ bandr) {
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
  BandR gr_[BAND ? kQT : 1];  // the same bands in the pair's frame, for the box tests
  if (BAND) {
#pragma unroll
    for (int qt = 0; qt < (BAND ? kQT : 1); ++qt) gr_[qt] = bandr[qbase + qt * 32 + col];
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
#if SSRLCV_MATCH_F16_ORDER
    __builtin_amdgcn_sched_barrier(0);  // (chains 0 and 1 are independent: without this the scheduler interleaves them)
#endif
#pragma unroll
    for (int qt = 1; qt < kQT; ++qt) {
      floatx16& cur = (qt & 1) ? acc1 : acc0;
      floatx16& prev = (qt & 1) ? acc0 : acc1;
      cur = floatx16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < kKSteps; ++s) cur = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], bq[qt][s], cur, 0, 0, 0);
#if SSRLCV_MATCH_F16_ORDER
      // chain qt stays complete in the instruction stream in front of the epilogue of chain qt - 1 (whose inline-asm reads the
      // hazard recogniser does not see): an interleaving of two chains would read an accumulator one MFMA behind its last write
      __builtin_amdgcn_sched_barrier(0);
#endif
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
      for (int qt = 0; qt < (BAND ? kQT : 1); ++qt) gh = gh || band_hits_box(gr_[qt], gb);
      if (!__any(gh)) continue;
      uint32_t t1 = (gr + 1) * 32;
      if (t1 > tile1) t1 = tile1;
      for (uint32_t tt = gr * 32 > tile0 ? gr * 32 : tile0; tt < t1; ++tt) {
        const Box tb = tileBox[tt];
        unsigned need = 0;
#pragma unroll
        for (int qt = 0; qt < (BAND ? kQT : 1); ++qt)
          if (__any(band_hits_box(gr_[qt], tb))) need |= 1u << qt;
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
  // two waves per SIMD: the partner wave hides the load latency, one register buffer suffices.  (Round 5 measured two ways of
  // having the next tile in flight -- LDS-DMA into a per-wave slot, and K step by K step behind the last chain -- both exact,
  // both slower: 15.3 / 14.3 ms against 13.2, profiles/r05_kernel_ab.txt; the code is in the history.)
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

// Which formulation of the distance contraction the matcher calls use (process-wide; ssrlcv_hip_set_match_arithmetic):
// both are exact, so the choice never changes a result.  The developer build starts from SSRLCV_MATCH_F16.
std::atomic<int>& match_arithmetic() {
  static std::atomic<int> a(svdev::env("SSRLCV_MATCH_F16") != nullptr ? SSRLCV_MATCH_ARITH_F16 : SSRLCV_MATCH_ARITH_I8);
  return a;
}

struct Layout {
  uint32_t nq_pad, nt_pad;
  size_t off_pq, off_pt, off_nq, off_nt, off_lt, off_geom, off_key, off_scratch, off_permq, off_permt, off_tilebox, off_groupbox,
      off_bandr, off_frame, off_tilepar, off_sort, sort_bytes, total;
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
  L.off_bandr = take((size_t)L.nq_pad * sizeof(BandR));
  L.off_frame = take(sizeof(Frame));
  L.off_tilepar = take((size_t)(L.nt_pad / 32) * 4);
  L.sort_bytes = svm::sort_scratch_bytes(nq > nt ? nq : nt);
  L.off_sort = take(L.sort_bytes);
  L.total = o;
  return L;
}

int run_match(const ssrlcv_sift_feature* query, uint32_t nq, const ssrlcv_sift_feature* target, uint32_t nt, int mode,
              const ssrlcv_match_params* p, float absThreshold, char* ws, const Layout& L, hipStream_t st, bool seedOnly = false) {
  _Float16* pq = (_Float16*)(ws + L.off_pq);
  _Float16* pt = (_Float16*)(ws + L.off_pt);
  float* nqv = (float*)(ws + L.off_nq);
  ssrlcv_float2* lt = (ssrlcv_float2*)(ws + L.off_lt);
  Geom* geom = (Geom*)(ws + L.off_geom);
  unsigned long long* keys = (unsigned long long*)(ws + L.off_key);
  const bool band = mode != 0;
  uint32_t* permQ = band ? (uint32_t*)(ws + L.off_permq) : nullptr;
  uint32_t* permT = band ? (uint32_t*)(ws + L.off_permt) : nullptr;
  Frame* frame = (Frame*)(ws + L.off_frame);
  BandR* bandr = (BandR*)(ws + L.off_bandr);
  float* F9 = (float*)(ws + L.off_scratch);
  if (mode == 2) SSRLCV_HIP_TRY(hipMemcpyAsync(F9, p->fundamental, 9 * sizeof(float), hipMemcpyHostToDevice, st));
  auto launch_geom = [&](const uint32_t* perm) {
    if (mode == 1)
      hipLaunchKernelGGL(k_geom, dim3((L.nq_pad + 255) / 256), dim3(256), 0, st, query, nq, L.nq_pad, perm, p->queryCamera,
                         p->targetProjection[0], p->targetProjection[1], p->targetProjection[2], p->epsilon, p->delta, geom);
    if (mode == 2)
      hipLaunchKernelGGL(k_geom_fundamental, dim3((L.nq_pad + 255) / 256), dim3(256), 0, st, query, nq, L.nq_pad, perm,
                         (const float*)F9, geom);
  };
  if (band) {
    // the frame of the pair and the two orders (see "band culling"): bands of the queries in caller order -> u ->
    // targets by (strip across u, position along u) -> queries by where their band lies
    static const float forceDeg = svdev::env("SSRLCV_BAND_DIR") ? (float)atof(svdev::env("SSRLCV_BAND_DIR")) : 1e30f;
    // strip width: 0.4 epsilon within [4, 16] px (the bands are 2 epsilon thick; end of round 4, epsilon 25 on the 4-view
    // 4096^2 flow: 6 px 6.48 ms, 8 6.57, 10 6.54, 12 6.58, 16 6.87, 20 7.43, 24 7.2; queries at half or twice the targets'
    // width no better)
    static const float stripEnvT = svdev::env("SSRLCV_BAND_STRIP") ? (float)atof(svdev::env("SSRLCV_BAND_STRIP")) : 0.0f;
    static const float stripEnvQ = svdev::env("SSRLCV_BAND_STRIP_Q") ? (float)atof(svdev::env("SSRLCV_BAND_STRIP_Q")) : 0.0f;
    const float epsAbs = fabsf(p->epsilon);
    const float stripAuto = !(epsAbs == epsAbs) ? (float)SSRLCV_BAND_STRIP : fminf(16.0f, fmaxf(4.0f, 0.4f * epsAbs));
    const float stripT = stripEnvT > 0.0f ? stripEnvT : stripAuto;
    const float stripQ = stripEnvQ > 0.0f ? stripEnvQ : stripT;
    launch_geom(nullptr);
    hipLaunchKernelGGL(k_band_direction, dim3(1), dim3(1024), 0, st, (const Geom*)geom, nq, mode, forceDeg, frame);
    if (nt) {
      const svm::SortBuffers sb = svm::sort_buffers(ws + L.off_sort, nt);
      hipLaunchKernelGGL(k_target_keys, dim3((nt + 255) / 256), dim3(256), 0, st, target, nt, frame, 1.0f / stripT,
                         mode == 2 ? 1 : 0, sb.keys);
      int rc = svm::sort_filled_keys(nt, permT, ws + L.off_sort, L.sort_bytes, st);
      if (rc) return rc;
    }
    if (nq) {
      const svm::SortBuffers sb = svm::sort_buffers(ws + L.off_sort, nq);
      hipLaunchKernelGGL(k_query_keys, dim3((nq + 255) / 256), dim3(256), 0, st, (const Geom*)geom, nq, mode, p->epsilon,
                         (const Frame*)frame, 1.0f / stripQ, sb.keys);
      int rc = svm::sort_filled_keys(nq, permQ, ws + L.off_sort, L.sort_bytes, st);
      if (rc) return rc;
    }
  }
  // Integer formulation (matcher_i8.inc) by default; ssrlcv_hip_set_match_arithmetic(SSRLCV_MATCH_ARITH_F16) -- or, in the
  // developer build, SSRLCV_MATCH_F16=1 -- selects the fp16 one (same results).
  const bool useF16 = match_arithmetic().load(std::memory_order_relaxed) == SSRLCV_MATCH_ARITH_F16;
  if (useF16) {
    hipLaunchKernelGGL(k_pack, dim3((L.nq_pad * 16 + 255) / 256), dim3(256), 0, st, query, nq, L.nq_pad, 0, permQ, pq, nqv,
                       (ssrlcv_float2*)nullptr);
    hipLaunchKernelGGL(k_pack, dim3((L.nt_pad * 16 + 255) / 256), dim3(256), 0, st, target, nt, L.nt_pad, 1, permT, pt,
                       (float*)nullptr, lt);
  } else {
    hipLaunchKernelGGL(k_pack_i8, dim3((L.nq_pad * 16 + 255) / 256), dim3(256), 0, st, query, nq, L.nq_pad, 0, permQ,
                       (uint8_t*)pq, (int*)nqv, (ssrlcv_float2*)nullptr, 0);
    // targets: tile records in the order of the matchers' LDS slots (the region is sized for the fp16 rows: 9216 bytes
    // per tile, a record takes 5120)
    hipLaunchKernelGGL(k_pack_i8, dim3((L.nt_pad * 16 + 255) / 256), dim3(256), 0, st, target, nt, L.nt_pad, 1, permT,
                       (uint8_t*)pt, (int*)(ws + L.off_nt), lt, 1);
  }
  SSRLCV_HIP_TRY(hipMemsetAsync(keys, 0xff, (size_t)L.nq_pad * 8, st));
  const float eps = mode != 0 ? p->epsilon : 0.0f;
  launch_geom(permQ);  // (packed order; no-op in mode 0)
  if (band)
    hipLaunchKernelGGL(k_bandr, dim3((L.nq_pad + 255) / 256), dim3(256), 0, st, (const Geom*)geom, nq, L.nq_pad, mode, eps,
                       (const Frame*)frame, bandr);
  uint32_t qblocks = L.nq_pad / (useF16 ? kQPerBlock : kQPerBlock8);
  uint32_t numTiles = L.nt_pad / 32;
  const int* ntn = (const int*)(ws + L.off_nt);
  if (band) {
    Box* tileBox = (Box*)(ws + L.off_tilebox);
    Box* groupBox = (Box*)(ws + L.off_groupbox);
    const uint32_t numGroups = (numTiles + 31) / 32;
    uint32_t* tilePar = (uint32_t*)(ws + L.off_tilepar);
    hipLaunchKernelGGL(k_tile_boxes, dim3((numTiles + 255) / 256), dim3(256), 0, st, lt, nt, numTiles, (const Frame*)frame,
                       tileBox, useF16 ? (const int*)nullptr : ntn, useF16 ? (uint32_t*)nullptr : tilePar);
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
                         tilesPerSplit, mode, eps, absThreshold, keys, permT, tileBox, groupBox, (const BandR*)bandr);
    else
      hipLaunchKernelGGL((k_match_i8<true, kQT8Band>), dim3(qbBand, splits), dim3(256), 0, st, (const uint8_t*)pq, (const uint8_t*)pt,
                         (const int*)nqv, ntn, lt, geom, nq, nt, tilesPerSplit, mode, eps, absThreshold, keys, permT,
                         tileBox, groupBox, (const BandR*)bandr, (const uint32_t*)tilePar);
  } else {
    // split the target range until the grid has >= 1024 blocks (4 per CU) or tiles run out
    uint32_t splits = 1;
    while (qblocks * splits < 1024 && splits * 2 <= numTiles && numTiles / (splits * 2) >= 16) splits *= 2;
    uint32_t tilesPerSplit = (numTiles + splits - 1) / splits;
    if (useF16)
      hipLaunchKernelGGL(k_match<false>, dim3(qblocks, splits), dim3(256), 0, st, pq, pt, nqv, lt, geom, nq, nt,
                         tilesPerSplit, mode, eps, absThreshold, keys, (const uint32_t*)nullptr, (const Box*)nullptr,
                         (const Box*)nullptr, (const BandR*)nullptr);
    else if (seedOnly)  // minimum distance only (getSeedDistances)
      hipLaunchKernelGGL((k_match_i8<false, kQT8Brute, true>), dim3(qblocks, splits), dim3(256), 0, st, (const uint8_t*)pq,
                         (const uint8_t*)pt, (const int*)nqv, ntn, lt, geom, nq, nt, tilesPerSplit, mode, eps, absThreshold, keys,
                         (const uint32_t*)nullptr, (const Box*)nullptr, (const Box*)nullptr, (const BandR*)nullptr,
                         (const uint32_t*)nullptr);
    else
      hipLaunchKernelGGL((k_match_i8<false, kQT8Brute>), dim3(qblocks, splits), dim3(256), 0, st, (const uint8_t*)pq, (const uint8_t*)pt,
                         (const int*)nqv, ntn, lt, geom, nq, nt, tilesPerSplit, mode, eps, absThreshold, keys,
                         (const uint32_t*)nullptr, (const Box*)nullptr, (const Box*)nullptr, (const BandR*)nullptr,
                         (const uint32_t*)nullptr);
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

int ssrlcv_hip_set_match_arithmetic(int arithmetic) {
  if (arithmetic != SSRLCV_MATCH_ARITH_I8 && arithmetic != SSRLCV_MATCH_ARITH_F16) return SSRLCV_ERR_INVALID_ARG;
  match_arithmetic().store(arithmetic, std::memory_order_relaxed);
  return SSRLCV_OK;
}
int ssrlcv_hip_get_match_arithmetic(void) { return match_arithmetic().load(std::memory_order_relaxed); }

size_t ssrlcv_hip_match_workspace_bytes(uint32_t numQuery, uint32_t numTarget) {
  return make_layout(numQuery, numTarget).total;
}

#ifdef SSRLCV_MATCH_STATS
// developer builds with -DSSRLCV_MATCH_STATS only (not declared in the header): reads and clears the walk counters
int ssrlcv_dbg_match_stats(unsigned long long* out12) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out12, HIP_SYMBOL(g_match_stats), 12 * sizeof(unsigned long long));
  unsigned long long z[12] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_match_stats), z, sizeof(z));
  float f[8];
  (void)hipMemcpyFromSymbol(f, HIP_SYMBOL(g_match_fstats), sizeof(f));
  if (f[4] > 0.0f)
    printf("bands per wave: spread of wc %.1f px, of vc %.1f px; mean half-length %.1f px, half-thickness %.1f px; %.0f waves, %.0f lanes without a finite band\n",
           f[0] / f[4], f[1] / f[4], f[2] / f[4], f[3] / f[4], f[4], f[5]);
  float zf[8] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_match_fstats), zf, sizeof(zf));
  unsigned long long cy[8];
  (void)hipMemcpyFromSymbol(cy, HIP_SYMBOL(g_match_cycles), sizeof(cy));
  if (cy[5])
    printf("s_memtime of the band-culled waves: whole 100 %%; waiting for the tile in flight %.1f %%, slot -> registers %.1f %%, queueing the next "
           "transfer %.1f %%, chains + epilogue %.1f %% (of which the slow path %.1f %%), walk between hits %.1f %%\n",
           100.0 * cy[0] / cy[5], 100.0 * cy[1] / cy[5], 100.0 * cy[2] / cy[5], 100.0 * cy[3] / cy[5], 100.0 * cy[6] / cy[5],
           100.0 * cy[4] / cy[5]);
  unsigned long long zc[8] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_match_cycles), zc, sizeof(zc));
  return 0;
}
#endif

int ssrlcv_hip_seed_distances_u8x128(const ssrlcv_sift_feature* query, uint32_t numQuery, const ssrlcv_sift_feature* seed,
                                     uint32_t numSeed, float* out, void* workspace, size_t workspaceBytes,
                                     ssrlcv_stream_t stream) {
  if (!query || !out || !workspace || (numSeed && !seed)) return SSRLCV_ERR_INVALID_ARG;
  if (numQuery == 0) return SSRLCV_OK;
  Layout L = make_layout(numQuery, numSeed);
  if (workspaceBytes < L.total) return SSRLCV_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int rc = run_match(query, numQuery, seed, numSeed, 0, nullptr, FLT_MAX, (char*)workspace, L, st, true);
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
  if (reinterpret_cast<size_t>(keyPoints) & 15) return SSRLCV_ERR_INVALID_ARG;  // elements are written with one 16-byte store
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
