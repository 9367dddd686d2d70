// ssrlcv_amd/csrc/keypoints.hip -- key-point detection, refinement, orientation and 128-D descriptors for gfx950
// (SURVEY.md section 8a rows S8-S14).
//
// Everything after the pyramid works on small lists gathered from the raw DoG levels; nothing is copied to the host
// (the reference synchronises and round-trips counts after every launch).  Device-resident OctaveState mirrors
// Octave::extrema / extremaBlurIndices (include/FeatureFactory.cuh:107-123); single-thread bookkeeping kernels replay
// the reference's host-side index arithmetic literally, including the blur re-scan of refineExtremaLocation
// (src/FeatureFactory.cu:251-259) whose untouched entries keep the values discardExtrema left behind.
//
// Neither normalisation of the reference is materialised, and neither are the DoG levels: the workspace holds the six
// un-normalised Gaussian levels of every octave with their {min, max}, and the {min, max} of the five DoG levels
// (ssrlcv_hip_sift_build_dog reduces them in the pass that also finds the extrema).  A consumer that needs the
// twice-normalised DoG value of a pixel (findKeyPoints normalises the DoG levels again, src/FeatureFactory.cu:472)
// evaluates ((N(level b+1) - N(level b)) - min_b) / (max_b - min_b) with N(x) = (x - min) / (max - min): the float
// operations of normalize (src/Image.cu:1560-1565) and subtractImages (:842-845) in their order, so the values are the
// reference's bit for bit while 20 + 40 bytes per pixel of DoG traffic never exist.
// Gradients (calculatePixelGradients, src/Image.cu:1583-1598) are evaluated once per pixel into the polar tables.
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include "compact.h"
#include "dev_switch.h"
#include "scan_lookback.h"
#include "device_math.h"
#include "sift_plan.h"
#include "ssrlcv_hip.h"

using svp::OctaveState;

namespace {

struct LevelSet {  // one octave's scale space
  const float* lvl[svp::kGauss];  // un-normalised Gaussian levels
  const float* lvlMinMax;         // 6 x {min, max} of them
  const float* minmax;            // 5 x {min, max} of the raw DoG levels
  const float2* polar;            // {|grad|, atan2(gy,gx)} of the normalised DoG levels 1..3, level-major
  int w, h;
};

// DoG level b of an octave as its consumers see it: the two Gaussian levels it is the difference of and the constants
// of the three normalisations.  sv::div_by returns the IEEE quotient through a shared reciprocal (device_math.h).
struct DogView {
  const float* lo;
  const float* hi;
  float mnLo, mnHi, mnDog;
  sv::Divisor rgLo, rgHi, rgDog;
  __device__ __forceinline__ float raw(size_t a) const {  // subtractImages of the two normalised levels: k_dog's arithmetic
    return sv::div_by(hi[a] - mnHi, rgHi) - sv::div_by(lo[a] - mnLo, rgLo);
  }
  __device__ __forceinline__ float norm(size_t a) const { return sv::div_by(raw(a) - mnDog, rgDog); }
};
__device__ __forceinline__ DogView dog_view(const LevelSet& L, int b) {
  DogView v;
  v.lo = L.lvl[b];
  v.hi = L.lvl[b + 1];
  v.mnLo = L.lvlMinMax[2 * b];
  v.rgLo = sv::make_divisor(L.lvlMinMax[2 * b + 1] - v.mnLo);
  v.mnHi = L.lvlMinMax[2 * b + 2];
  v.rgHi = sv::make_divisor(L.lvlMinMax[2 * b + 3] - v.mnHi);
  v.mnDog = L.minmax[2 * b];
  v.rgDog = sv::make_divisor(L.minmax[2 * b + 1] - v.mnDog);
  return v;
}

// Where a kernel takes its DoG values from: the plan's workspace (Gaussian levels + constants, evaluated per sample) or
// -- the per-kernel exports over a caller's own ScaleSpace (round 4, end of this file) -- materialised DoG images as
// upstream holds them in Octave::blurs[b]->pixels at that point of its flow.
struct PlanSource {
  LevelSet L;
  __device__ __forceinline__ DogView view(int b) const { return dog_view(L, b); }
};
struct MatView {
  const float* p;
  __device__ __forceinline__ float norm(size_t a) const { return p[a]; }
};
struct MatSource {
  const float* const* px;  // device array of the octave's DoG level pointers
  __device__ __forceinline__ MatView view(int b) const { return MatView{px[b]}; }
};

// flagEdges' test on one key point (src/FeatureFactory.cu:974-990): D = the view of the level it reads
template <typename View>
__device__ __forceinline__ bool edge_response_above(const View& D, const ssrlcv_sskeypoint& kp, int W, float thr) {
  const int lx = (int)roundf(kp.loc.x), ly = (int)roundf(kp.loc.y);
#define NS(yy, xx) D.norm((size_t)(yy) * W + (xx))
  float h00 = -2.0f * NS(ly, lx);
  float h11 = h00 + NS(ly + 1, lx) + NS(ly - 1, lx);
  h00 += NS(ly, lx + 1) + NS(ly, lx - 1);
  float h01 = (NS(ly + 1, lx + 1) - NS(ly - 1, lx + 1) - NS(ly + 1, lx - 1) + NS(ly - 1, lx - 1));
#undef NS
  float e = h00 + h11;
  float det = (h00 * h11) - (h01 * h01);
  return (e * e / det) > thr;
}

// ---- bookkeeping (one thread: the last block of the partition that precedes it, see compact.h last_block_post) ----------
__global__ void k_state_reset(OctaveState* st) {
  for (int i = 0; i < svp::kDog; ++i) { st->idx[i] = 0; st->stale[i] = 0; }
  st->n = 0;
  st->hasExtrema = 0;
  st->overflow = 0;
}
// after searchForExtrema (src/FeatureFactory.cu:98-151): totals = counts for b = 1,2,3
__device__ __forceinline__ void book_extrema(OctaveState* st, const uint32_t* totals, uint32_t cap) {
  uint32_t c1 = totals[0], c2 = totals[1], total = totals[3];
  // the list is grouped by blur (1, 2, 3) and the scatter drops everything past `cap`: on overflow the list is
  // truncated there (the flag is reported by ssrlcv_sift_plan_overflow; the reference's lists are unbounded)
  if (total > cap) {
    st->overflow = 1;
    total = cap;
    c1 = c1 < cap ? c1 : cap;
    c2 = c2 < cap - c1 ? c2 : cap - c1;
  }
  st->idx[0] = 0;
  st->idx[1] = 0;
  st->idx[2] = (int)c1;
  st->idx[3] = (int)(c1 + c2);
  if (total) st->idx[4] = (int)total;
  st->n = (int)total;
  st->hasExtrema = total != 0;
}
// after discardExtrema (src/FeatureFactory.cu:161-215): totals[s] = survivors of segment s
__device__ __forceinline__ void book_discard(OctaveState* st, const uint32_t* totals) {
  if (!st->hasExtrema) return;
  int kept = 0;
  for (int i = 0; i < svp::kDog; ++i) {
    st->idx[i] = kept;
    kept += (int)totals[i];
  }
  st->n = kept;
  if (kept == 0) st->hasExtrema = 0;
}
// stable_sort by blur + host re-scan of refineExtremaLocation (src/FeatureFactory.cu:249-259): totals[v] = survivors
// with blur == v.  Entries of idx the loop does not reach keep what k_book_discard wrote.
__device__ __forceinline__ void book_rescan(OctaveState* st, const uint32_t* totals) {
  if (!st->hasExtrema) return;
  st->idx[0] = 0;
  st->idx[1] = 0;
  int pos = 0, blur = 2;
  bool first = true;
  for (int v = 0; v < svp::kDog; ++v) {
    int c = (int)totals[v];
    if (c == 0) continue;
    if (!first && blur < svp::kDog - 1) st->idx[blur++] = pos;  // host[i-1] < host[i] at i = pos
    first = false;
    pos += c;
  }
  st->idx[svp::kDog - 1] = st->n;
}
// after computeKeyPointOrientations (src/FeatureFactory.cu:561-630): totals[s] = oriented key points of segment s
__global__ void k_book_extrema(OctaveState* st, const uint32_t* totals, uint32_t cap) { book_extrema(st, totals, cap); }
__device__ __forceinline__ void book_orient(OctaveState* st, const uint32_t* totals, uint32_t cap) {
  if (!st->hasExtrema) return;
  int total = 0;
  for (int b = 0; b < svp::kDog; ++b) {
    st->idx[b] = total < (int)cap ? total : (int)cap;
    total += (int)totals[b];
  }
  if ((uint32_t)total > cap) { st->overflow = 1; total = (int)cap; }  // truncated: the scatter drops what lies past cap
  st->n = total;
  if (total == 0) st->hasExtrema = 0;
}
__global__ void k_book_featbase(OctaveState* st, uint32_t* featBase, uint32_t* numFeatures, uint32_t maxFeatures) {
  uint32_t tot = 0;
  for (int o = 0; o < svp::kOctaves; ++o) {
    featBase[o] = tot;
    tot += st[o].hasExtrema ? (uint32_t)st[o].n : 0u;
  }
  if (tot > maxFeatures) tot = maxFeatures;
  *numFeatures = tot;
}

__device__ __forceinline__ int segment_of(const OctaveState* st, int i) {
  int s = 0;
#pragma unroll
  for (int k = 1; k < svp::kDog; ++k)
    if (st->idx[k] <= i) s = k;
  return s;
}

// ---- S9 / S11 / S12 flag kernels -----------------------------------------------------------------------------------------
// flagNoise (src/FeatureFactory.cu:968-973)
__global__ __launch_bounds__(256) void k_flag_noise(const OctaveState* st, ssrlcv_sskeypoint* kps, float thr) {
  int n = st->hasExtrema ? st->n : 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    kps[i].discard = (uint8_t)(fabsf(kps[i].intensity) < thr);
}

// flagEdges (src/FeatureFactory.cu:974-990) on the level of the key point's SEGMENT (removeEdges :287-306)
__global__ __launch_bounds__(256) void k_flag_edges(const OctaveState* st, ssrlcv_sskeypoint* kps, LevelSet L, float thr) {
  int n = st->hasExtrema ? st->n : 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const DogView D = dog_view(L, segment_of(st, i));
    kps[i].discard = (uint8_t)edge_response_above(D, kps[i], L.w, thr);
  }
}

// removeNoise + removeEdges + checkKeyPoints (src/SIFT_FeatureFactory.cu:449-461) in one pass.  The three tests are independent per key point and the
// compaction is stable, so one discard of their union leaves the list the three successive discards leave; used when
// the extraction runs past stage 5 (the staged kernels above serve the per-stage stop points of the tests).
__global__ __launch_bounds__(256) void k_flag_noise_edges_window(const OctaveState* st, ssrlcv_sskeypoint* kps, LevelSet L,
                                                                 float noiseThr, float edgeThr, float pixelWidth,
                                                                 float lambda) {
  int n = st->hasExtrema ? st->n : 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    ssrlcv_sskeypoint kp = kps[i];
    bool drop = fabsf(kp.intensity) < noiseThr;
    drop = drop || edge_response_above(dog_view(L, segment_of(st, i)), kp, L.w, edgeThr);
    float ww = kp.sigma * lambda / pixelWidth;
    drop = drop || (kp.loc.x - ww) < 0.0f || (kp.loc.y - ww) < 0.0f || (kp.loc.x + ww) >= (unsigned)(L.w - 1) ||
           (kp.loc.y + ww) >= (unsigned)(L.h - 1);
    kps[i].discard = (uint8_t)drop;
  }
}

// checkKeyPoints alone (src/SIFT_FeatureFactory.cu:449-461), for the stage-at-a-time entry point
__global__ __launch_bounds__(256) void k_flag_window(const OctaveState* st, ssrlcv_sskeypoint* kps, int w, int h, float pixelWidth,
                                                     float lambda) {
  int n = st->hasExtrema ? st->n : 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const ssrlcv_sskeypoint kp = kps[i];
    const float ww = kp.sigma * lambda / pixelWidth;
    kps[i].discard = (uint8_t)((kp.loc.x - ww) < 0.0f || (kp.loc.y - ww) < 0.0f || (kp.loc.x + ww) >= (unsigned)(w - 1) ||
                               (kp.loc.y + ww) >= (unsigned)(h - 1));
  }
}

// ---- S10: refineLocation (src/FeatureFactory.cu:892-967) -----------------------------------------------------------------
// refineLocation on one key point (src/FeatureFactory.cu:892-967); src.view(b) = DoG level b as it is at that point
template <typename Src>
__device__ __forceinline__ void refine_one(ssrlcv_sskeypoint& kp, int W, int H, int numBlurs, float sigmaMin, float mult, const Src& src) {
  int lx = (int)roundf(kp.loc.x), ly = (int)roundf(kp.loc.y);
  float hess[3][3], hinv[3][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
  float grad[3], temp[3], offset[3] = {0.0f, 0.0f, 0.0f};
  int bl = kp.blur;
  for (int attempt = 0; attempt < 5; ++attempt) {
    const auto Dl = src.view(bl - 1), Dm = src.view(bl), Du = src.view(bl + 1);
#define PM(yy, xx) Dm.norm((size_t)(yy) * W + (xx))
#define PL(yy, xx) Dl.norm((size_t)(yy) * W + (xx))
#define PU(yy, xx) Du.norm((size_t)(yy) * W + (xx))
    grad[0] = PM(ly, lx + 1) - PM(ly, lx - 1);
    grad[1] = PM(ly + 1, lx) - PM(ly - 1, lx);
    grad[2] = PU(ly, lx) - PL(ly, lx);
    float centre = PM(ly, lx);
    hess[0][0] = grad[0] - 2 * centre;
    hess[0][1] = (PM(ly + 1, lx + 1) - PM(ly - 1, lx + 1) - PM(ly + 1, lx - 1) + PM(ly - 1, lx - 1)) / 4.0f;
    hess[0][2] = (PU(ly, lx + 1) - PL(ly, lx + 1) - PU(ly, lx - 1) + PL(ly, lx - 1)) / 4.0f;
    hess[1][0] = hess[0][1];
    hess[1][1] = grad[1] - 2 * centre;
    hess[1][2] = (PU(ly + 1, lx) - PL(ly + 1, lx) - PU(ly - 1, lx) + PL(ly - 1, lx)) / 4.0f;
    hess[2][0] = hess[0][2];
    hess[2][1] = hess[1][2];
    hess[2][2] = grad[2] - 2 * centre;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) hess[r][c] *= -1.0f;
    sv::inverse3(hess, hinv);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float val = 0;
#pragma unroll
      for (int c = 0; c < 3; ++c) val = __builtin_fmaf(hinv[r][c], grad[c], val);  // multiply(): nvcc's fused chain (device_math.h)
      offset[r] = val;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float val = 0;
#pragma unroll
      for (int r = 0; r < 3; ++r) val = __builtin_fmaf(hess[r][c], grad[r], val);
      temp[c] = val;
    }
    if (fabsf(offset[0]) <= 0.5f && fabsf(offset[1]) <= 0.5f && fabsf(offset[2]) <= 0.5f) {
      kp.loc.x = (float)lx + offset[0];
      kp.loc.y = (float)ly + offset[1];
      lx = (int)roundf(kp.loc.x);
      ly = (int)roundf(kp.loc.y);
      kp.discard = (uint8_t)(lx <= 0 || ly <= 0 || lx >= W - 1 || ly >= H - 1);
      if (kp.discard) break;
      kp.intensity = PM(ly, lx) - (0.5f * ((temp[0] * grad[0]) + (temp[1] * grad[1]) + (temp[2] * grad[2])));
      kp.sigma = sigmaMin * sv_powf(mult, ((float)bl + offset[2]));
      if (fabsf(offset[2]) > 0.5) bl += (offset[2] > 0) ? 1 : -1;
      break;
    } else if (attempt == 4) {
      kp.discard = 1;
      break;
    } else {
      if (fabsf(offset[0]) > 0.5) lx += (offset[0] > 0) ? 1 : -1;
      if (fabsf(offset[1]) > 0.5) ly += (offset[1] > 0) ? 1 : -1;
      kp.loc.x = (float)lx;
      kp.loc.y = (float)ly;
      if (fabsf(offset[2]) > 0.5) bl += (offset[2] > 0) ? 1 : -1;
      if (bl >= numBlurs - 1 || bl <= 0 || lx <= 0 || ly <= 0 || lx >= W - 1 || ly >= H - 1) {
        kp.discard = 1;
        break;
      }
    }
#undef PM
#undef PL
#undef PU
  }
  kp.blur = bl;
}

__global__ __launch_bounds__(256) void k_refine(const OctaveState* st, ssrlcv_sskeypoint* kps, LevelSet L, float sigmaMin,
                                                float mult) {
  int n = st->hasExtrema ? st->n : 0;
  const PlanSource src{L};
  for (int gi = blockIdx.x * 256 + threadIdx.x; gi < n; gi += gridDim.x * 256) {
    ssrlcv_sskeypoint kp = kps[gi];
    refine_one(kp, L.w, L.h, svp::kDog, sigmaMin, mult, src);
    kps[gi] = kp;
  }
}

// Gradient magnitude / direction of every pixel of the normalised DoG levels 1..3, computed once per image: the
// orientation and descriptor windows of neighbouring key points overlap ~10x, and 4 gathers + 4 divisions + sqrtf +
// atan2f per sample was two thirds of their instruction count.  The operations are those of normalize + subtractImages +
// normalize + calculatePixelGradients, so values are bit-identical to the per-sample path.  A block owns a 256 x 16 tile
// and every thread a column run of it; it walks the three DoG levels of the tile in turn, keeping the normalised column of
// the upper Gaussian level in registers as the lower operand of the next DoG level, so the four Gaussian levels 1..4 are
// read once (16 B per pixel; the DoG levels are never in memory) and every Gaussian pixel is normalised once.  Vertical
// neighbours are the thread's own registers, horizontal ones come through LDS, and the border rule of
// calculatePixelGradients (src/Image.cu:1583-1598: a border pixel takes the stencil of its inner neighbour) is a choice
// of indices, not per-pixel address arithmetic.
constexpr int kPolRows = 16;
struct PolarJobs {  // the four octaves' tables in one launch: octave o owns the blocks start[o] .. start[o + 1] - 1
  LevelSet L[svp::kOctaves];
  float2* out[svp::kOctaves];
  uint32_t start[svp::kOctaves + 1];
  uint32_t tilesX[svp::kOctaves];
};
// k_polar streams: it reads four Gaussian levels once and writes 2.2 GB of tables that no cache holds until the sampling
// kernels gather them -- non-temporal loads and stores (describe 3.06-3.09 -> 2.95-3.04 ms per 4096^2 image; the stores
// alone 2.99-3.02).  The same hint on the Gaussian kernels' level stores gained 10-18 us per level alone and nothing
// inside build_dog: a plainly stored 268 MB level is still largely in the 256 MB memory-side cache when the next level
// reads it.  -DSSRLCV_NT_STORES=0 builds the plain form (A/B).
#ifndef SSRLCV_NT_STORES
#define SSRLCV_NT_STORES 1
#endif
typedef float f32x2p __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float polar_ld(const float* p) {
#if SSRLCV_NT_STORES
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
__global__ __launch_bounds__(256) void k_polar(PolarJobs jobs) {
  __shared__ float s_n[kPolRows][256 + 2];  // column c of the tile at [.][c + 1]; [.][0] / [.][257] = columns x0 - 1 / x0 + 256
  int o = 0;
#pragma unroll
  for (int k = 1; k < svp::kOctaves; ++k)
    if (blockIdx.x >= jobs.start[k]) o = k;  // block-uniform
  const LevelSet& L = jobs.L[o];
  float2* __restrict__ out = jobs.out[o];
  const uint32_t tile = blockIdx.x - jobs.start[o];
  const int tileX = (int)(tile % jobs.tilesX[o]), tileY = (int)(tile / jobs.tilesX[o]);
  const int W = L.w, H = L.h;
  const int x0 = tileX * 256, y0 = tileY * kPolRows;
  const int t = threadIdx.x, x = x0 + t;
  // own column, rows y0 - 1 .. y0 + 16; rows / columns clamped into the image are loaded but never used (see below)
  const int xc = x < W ? x : W - 1;
  // halo role: thread t < 32 also carries one pixel of column x0 - 1 (even t) or x0 + 256 (odd t), row y0 + t / 2
  const bool haloRole = t < 2 * kPolRows;
  const int hside = t & 1;
  int hx = hside ? x0 + 256 : x0 - 1;
  hx = hx < 0 ? 0 : (hx > W - 1 ? W - 1 : hx);
  int hy = y0 + (t >> 1);
  hy = hy > H - 1 ? H - 1 : hy;
  size_t rowOff[kPolRows + 2];
#pragma unroll
  for (int j = 0; j < kPolRows + 2; ++j) {
    int yy = y0 - 1 + j;
    yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
    rowOff[j] = (size_t)yy * W + xc;
  }
  const size_t haloOff = (size_t)hy * W + hx;
  // normalised column of Gaussian level 1 (the lower operand of DoG level 1)
  float nlo[kPolRows + 2], nloH = 0.0f;
  {
    const float mn = L.lvlMinMax[2];
    const sv::Divisor rg = sv::make_divisor(L.lvlMinMax[3] - mn);
    const float* __restrict__ g = L.lvl[1];
#pragma unroll
    for (int j = 0; j < kPolRows + 2; ++j) nlo[j] = sv::div_by(polar_ld(g + rowOff[j]) - mn, rg);
    if (haloRole) nloH = sv::div_by(g[haloOff] - mn, rg);
  }
  // LDS columns of the two horizontal taps: x - 1 / x + 1, at the image border x / x + 2 resp. x - 2 / x
  const int cl = x == 0 ? 1 : (x == W - 1 ? t - 1 : t);
  const int cr = x == 0 ? 3 : (x == W - 1 ? t + 1 : t + 2);
#pragma unroll 1
  for (int lvl = 1; lvl <= 3; ++lvl) {
    const float mn = L.lvlMinMax[2 * (lvl + 1)];
    const sv::Divisor rg = sv::make_divisor(L.lvlMinMax[2 * (lvl + 1) + 1] - mn);
    const float dmn = L.minmax[2 * lvl];
    const sv::Divisor drg = sv::make_divisor(L.minmax[2 * lvl + 1] - dmn);
    const float* __restrict__ g = L.lvl[lvl + 1];
    float v[kPolRows + 2];
#pragma unroll
    for (int j = 0; j < kPolRows + 2; ++j) {
      const float nhi = sv::div_by(polar_ld(g + rowOff[j]) - mn, rg);
      v[j] = sv::div_by((nhi - nlo[j]) - dmn, drg);  // the twice-normalised DoG value
      nlo[j] = nhi;
    }
    if (lvl > 1) __syncthreads();  // the previous level's horizontal taps have been read
#pragma unroll
    for (int i = 0; i < kPolRows; ++i) s_n[i][t + 1] = v[i + 1];
    if (haloRole) {
      const float nhi = sv::div_by(g[haloOff] - mn, rg);
      s_n[t >> 1][hside ? 257 : 0] = sv::div_by((nhi - nloH) - dmn, drg);
      nloH = nhi;
    }
    __syncthreads();
    float2* __restrict__ lvlOut = out + (size_t)(lvl - 1) * svp::polar_level_stride(W, H);
    if (tile == 0) {  // the zero entries around the table (see svp::polar_level_stride)
      if (t == 0) lvlOut[0] = make_float2(0.0f, 0.0f);
      for (int i = t; i < W + 1; i += 256) lvlOut[1 + (size_t)W * H + i] = make_float2(0.0f, 0.0f);
    }
    if (x < W) {
      float2* __restrict__ o = lvlOut + 1 + (size_t)y0 * W + x;
#pragma unroll
      for (int i = 0; i < kPolRows; ++i) {
        const int y = y0 + i;
        if (y >= H) break;
        float2 g2;
        g2.x = s_n[i][cr] - s_n[i][cl];
        // rows y + 1 / y - 1 are v[i + 2] / v[i]; at the border rows y + 2 / y resp. y / y - 2
        const float up = y == 0 ? v[i + 3 < kPolRows + 2 ? i + 3 : kPolRows + 1] : (y == H - 1 ? v[i + 1] : v[i + 2]);
        const float dn = y == 0 ? v[i + 1] : (y == H - 1 ? v[i > 0 ? i - 1 : 0] : v[i]);
        g2.y = up - dn;
        float2 r;
        r.x = sqrtf((g2.x * g2.x) + (g2.y * g2.y));
        r.y = sv_atan2f(g2.y, g2.x);
#if SSRLCV_NT_STORES
        __builtin_nontemporal_store(f32x2p{r.x, r.y}, reinterpret_cast<f32x2p*>(o + (size_t)i * W));  // 2.2 GB per image, gathered later
#else
        o[(size_t)i * W] = r;
#endif
      }
    }
  }
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
// roundf / llroundf of a non-negative coordinate: v_cvt_rpi_i32_f32 = floor(v + 0.5) evaluated exactly (see round_coord)
__device__ __forceinline__ int round_pos(float v) {
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(v));
  return r;
}

// fmodf(v, p) for -p < v < 2p, exact: fmod is an exact operation and v - p is exact for p <= v < 2p (Sterbenz).  That
// range is all the sampling kernels feed it: atan2 in [-pi, pi], theta in [0, 2 pi), plus 2 pi.
__device__ __forceinline__ float fmod_2pi_above(float v, float p) { return v >= p ? v - p : v; }
// pixel (x, y) of a polar table by its flat index y * W + x (-1 .. W*H + W: see svp::polar_level_stride; `pl` points at
// flat index 0): the index fits 32 bits for any level up to 32768^2, which keeps the multiply 32-bit
__device__ __forceinline__ float2 polar_px(const float2* __restrict__ pl, int W, int x, int y) {
  return pl[y * W + x];
}
// llroundf of a window coordinate v > -1: v_cvt_rpi_i32_f32 is floor(v + 0.5) evaluated exactly (tools/f64_rate.hip:
// equal to round-half-up for every float in (-1, 2^31)), which is llroundf except at v == -0.5 (half away from zero)
__device__ __forceinline__ int round_coord(float v) {
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(v));
  return v == -0.5f ? -1 : r;
}

// ---- S13: computeThetas(SSKeyPoint) (src/FeatureFactory.cu:1004-1112) -----------------------------------------------------
// The reference runs ONE THREAD per key point and accumulates its 36-bin histogram with a sequential float
// multiply-add chain in raster order of the window (:1031-1047): that order is part of the result (float addition does
// not associate), and the peak tests and the parabolic interpolation read the histogram's last bits.  So this kernel
// keeps the reference's sum -- its two float loop counters, one fmaf per sample and bin in raster order -- and spends its
// effort around that chain:
//   - the histogram lives in LDS, one column per key point (bin-major), not in scratch;
//   - gradient magnitude / direction come from the per-pixel polar tables (k_polar) instead of 4 gathers + sqrtf +
//     atan2f per sample;
//   - 1 << LPKS lanes share a key point (four since round 6): a chunk is 8 << LPKS consecutive samples of a window row,
//     fetched as neighbouring 16-byte pieces (the next chunk requested before the current one is processed), every lane
//     evaluates the Gaussian weight and the bin of eight of them -- the expensive part, which no order constrains -- and
//     then the votes are added to the one histogram column lane after lane, in raster order.  With one lane per key point
//     (rounds 2-5) the kernel took as long as ONE lane's chain over the largest window (3 481 samples, 0.4 ms) while
//     the vector pipe sat at a quarter of its rate;
//   - the divisions are the IEEE quotients by way of sv::exact_div3 / exact_div5 (reciprocals hoisted), the Gaussian is
//     the shared expf;
//   - one-wave blocks: the windows of the key points of a wave differ in size and the wave runs as long as its largest;
//     many small blocks let the dispatcher even that out.
// The first version of this file ran one WAVE per key point with exact 64-bit fixed-point LDS atomics -- order
// independent, but not the reference's sum: thetas agreed to ~1e-6 and a handful of peak decisions per million key
// points flipped.
constexpr int kThetaChunk = 8;

// Work list of the two sampling kernels: the key points of ALL octaves in one launch.  Every octave's list is cut at
// its blur-segment boundaries (a segment = one DoG level = one polar table = one class of window sizes: windows grow
// with the blur index and are the same in every octave, sigma / pixelWidth being the same), and the ranges are laid
// out largest windows first.  One launch instead of four: the short octaves hold a few thousand key points, i.e. a
// handful of waves whose run time is that of ONE wave (0.4 ms for a blur-3 window) -- four launches in a row paid that
// latency four times (1.96 ms of orientation kernels for 216 k key points, 1.3 ms of it for 10 % of them).
struct RangeTable {
  uint32_t first[20], count[20], start[20];  // range r = order * 4 + octave: list index, length, first work unit
  uint32_t total;                            // work units in all ranges
  uint32_t pad[3];
};
static_assert(sizeof(RangeTable) == 256, "tables sit at 256-byte strides in the plan's group block");
__device__ __constant__ const int kSegOrder[5] = {3, 2, 1, 4, 0};
// unitShift: log2 of the key points per work unit (6: one 64-lane block of k_thetas; 0: one wave of k_descriptors)
// sel: bit o * 5 + seg = the range of blur segment `seg` of octave o is wanted (the others stay empty)
__device__ __forceinline__ void build_ranges(const OctaveState* st, RangeTable* tab, int unitShift, uint32_t sel) {
  uint32_t pos = 0;
  for (int k = 0; k < 5; ++k) {
    const int seg = kSegOrder[k];
    for (int o = 0; o < svp::kOctaves; ++o) {
      const int r = k * 4 + o;
      uint32_t first = 0, count = 0;
      if (st[o].hasExtrema && ((sel >> (o * 5 + seg)) & 1u)) {
        const int lo = st[o].idx[seg], hi = seg < svp::kDog - 1 ? st[o].idx[seg + 1] : st[o].n;
        if (hi > lo) { first = (uint32_t)lo; count = (uint32_t)(hi - lo); }
      }
      tab->first[r] = first;
      tab->count[r] = count;
      tab->start[r] = pos;
      pos += (count + (1u << unitShift) - 1) >> unitShift;
    }
  }
  tab->total = pos;
}
__global__ void k_build_ranges(const OctaveState* st, RangeTable* tab, int unitShift, uint32_t sel) { build_ranges(st, tab, unitShift, sel); }
// k_book_featbase + the descriptor work list of all octaves in one single-thread launch (two launches were ~10 us of gap)
__global__ void k_book_and_desc_ranges(OctaveState* st, uint32_t* featBase, uint32_t* numFeatures, uint32_t maxFeatures, RangeTable* tab) {
  uint32_t tot = 0;
  for (int o = 0; o < svp::kOctaves; ++o) {
    featBase[o] = tot;
    tot += st[o].hasExtrema ? (uint32_t)st[o].n : 0u;
  }
  if (tot > maxFeatures) tot = maxFeatures;
  *numFeatures = tot;
  build_ranges(st, tab, 0, 0xFFFFFu);
}

// ---- the sampling kernels in groups (round 5) ---------------------------------------------------------------------------------
// k_thetas is a latency chain per lane (VALU at a quarter of its issue rate), k_descriptors is bound by vector-instruction
// issue, and the second needs the first's result only per key point: the key points are cut into kSampleGroups groups that
// are CONTIGUOUS IN THE OUTPUT ORDER (octave-major, blur segment, raster) --
//     0: octave 0, segments 0-1     1: octave 0, segment 2     2: octave 0, segments 3-4     3: octaves 1-3
// -- so that a group's features can be written as soon as the groups in front of it are expanded (their counts are the
// offsets), and the orientation launches of the later groups run beside the descriptor launches of the earlier ones.
// Octave 0's three groups are pieces of ONE order-preserving expansion (k_expand_orient over an element range, each piece
// starting at the running total the piece before it left).  The cut points are extremaBlurIndices; the reference's blur
// re-scan can leave stale entries in them (book_rescan), so the device checks that they are an ordered partition of the
// list (`regular`); if not, group 0 takes the whole octave's orientations and group 2 all its descriptors -- the single
// launch order of rounds 2-4 -- and nothing depends on where the cuts fall.
constexpr int kOct0Groups = svp::kSampleGroups - 1;
struct GroupCtl {                       // one per octave, in the plan's group block (zeroed before the first expansion)
  uint32_t segStart[svp::kDog];         // offset + 1 of the first kept element of every segment (0 = not met yet)
  uint32_t totals[svp::kSampleGroups + 1];  // totals[g] = kept elements in front of piece g (totals[0] = 0)
  uint32_t groupSum[svp::kSampleGroups];    // kept elements of piece g (written by the block of its last tile)
  uint32_t done[svp::kSampleGroups];        // blocks of piece g's launch that have left
  uint32_t regular;                     // the octave's blur indices are an ordered partition of its list
  uint32_t cut[2];                      // list indices where pieces 1 and 2 start (n, n when not regular)
  uint32_t pad[32 - svp::kDog - 3 * svp::kSampleGroups - 4];
};
static_assert(sizeof(GroupCtl) == 128, "four of them + eight range tables fit the plan's 4 KB group block");
__device__ __forceinline__ bool blur_indices_regular(const OctaveState* st) {
  if (!st->hasExtrema) return true;
  bool ok = st->idx[0] == 0;
  for (int k = 1; k < svp::kDog; ++k) ok = ok && st->idx[k - 1] <= st->idx[k];
  return ok && st->idx[svp::kDog - 1] <= st->n;
}
// before the orientation launches: the four groups' orientation work lists and octave 0's cut points (one thread)
__global__ void k_build_group_ranges(const OctaveState* st, RangeTable* tabs, GroupCtl* ctl, int unitShift, bool forceIrregular) {
  const bool reg = blur_indices_regular(st) && !forceIrregular;
  const uint32_t n0 = st[0].hasExtrema && st[0].n > 0 ? (uint32_t)st[0].n : 0u;
  ctl[0].regular = reg ? 1u : 0u;
  ctl[0].cut[0] = reg && n0 ? (uint32_t)st[0].idx[2] : n0;
  ctl[0].cut[1] = reg && n0 ? (uint32_t)st[0].idx[3] : n0;
  const uint32_t oct0 = 0x1Fu, rest = 0xFFFFFu & ~oct0;
  build_ranges(st, tabs + 0, unitShift, reg ? 0x03u : oct0);
  build_ranges(st, tabs + 1, unitShift, reg ? 0x04u : 0u);
  build_ranges(st, tabs + 2, unitShift, reg ? 0x18u : 0u);
  build_ranges(st, tabs + 3, unitShift, rest);
}
// after piece g of octave 0's expansion: the descriptor work list of group g over the EXPANDED list.  Regular: the
// segments of the piece, from the running offsets the expansion recorded; not regular: nothing for groups 0 and 1, the
// whole octave (its final state, piece 2 has written it) for group 2.
__global__ void k_build_desc_ranges_oct0(const OctaveState* st, RangeTable* tab, const GroupCtl* ctl, int g, uint32_t cap) {
  if (!ctl->regular) {
    build_ranges(st, tab, 0, g == kOct0Groups - 1 ? 0x1Fu : 0u);
    return;
  }
  const int segLo = g == 0 ? 0 : g + 1, segHi = g == 0 ? 2 : (g == 1 ? 3 : svp::kDog);
  const uint32_t end = ctl->totals[g + 1] < cap ? ctl->totals[g + 1] : cap;
  auto start_of = [&](int seg) {  // a segment nobody has met yet is empty and lies at the end of what is expanded so far
    if (seg >= segHi) return end;
    const uint32_t enc = ctl->segStart[seg];
    const uint32_t v = enc ? enc - 1u : end;
    return v < end ? v : end;
  };
  uint32_t pos = 0;
  for (int k = 0; k < 5; ++k) {
    const int seg = kSegOrder[k];
    for (int o = 0; o < svp::kOctaves; ++o) {
      const int r = k * 4 + o;
      uint32_t first = 0, count = 0;
      if (o == 0 && seg >= segLo && seg < segHi) {
        const uint32_t lo = start_of(seg), hi = start_of(seg + 1);
        if (hi > lo) { first = lo; count = hi - lo; }
      }
      tab->first[r] = first;
      tab->count[r] = count;
      tab->start[r] = pos;
      pos += count;
    }
  }
  tab->total = pos;
}
// work unit u -> range (wave-uniform scan of the 20 starts; empty ranges share their successor's start and are skipped)
__device__ __forceinline__ int range_of(const RangeTable* tab, uint32_t u) {
  int r = 0;
#pragma unroll
  for (int k = 1; k < 20; ++k)
    if (tab->start[k] <= u) r = k;
  return r;
}

struct OctaveSet {  // per-octave arguments of the combined sampling kernels
  LevelSet L[svp::kOctaves];
  const ssrlcv_sskeypoint* kps[svp::kOctaves];
  float pixelWidth[svp::kOctaves];
  float* thetas[svp::kOctaves];
  uint32_t* thetaCnt[svp::kOctaves];
  const void* consts[svp::kOctaves];
};

template <int MAXO, int LPKS>
__global__ __launch_bounds__(64) void k_thetas(const OctaveState* states, const RangeTable* tab, OctaveSet set, float lambda,
                                               float orientationThreshold) {
  // LPKS: log2 of the lanes that share a key point (round 6).  The lanes of a group walk the window together, a chunk
  // being 8 << LPKS samples of a row: every lane weighs eight of them (the expensive, order-free part), then the votes
  // enter the group's ONE histogram column in raster order, lane after lane under the execution mask (LDS executes a
  // wave's instructions in order).  The chain a key point's run time hangs on is 1 << LPKS times shorter, and the lanes
  // of a group fetch neighbouring 16-byte pieces (one 16 << LPKS byte run per instruction instead of 1 << LPKS runs).
  constexpr int LANES = 1 << LPKS, PER_BLOCK = 64 >> LPKS, CW = kThetaChunk * LANES;
  __shared__ float s_hist[36][PER_BLOCK];  // one column per key point: the lanes that vote at the same time sit in different banks
  if (blockIdx.x >= tab->total) return;  // block-uniform
  const int range = __builtin_amdgcn_readfirstlane(range_of(tab, blockIdx.x));
  const int octave = range & 3, useg = kSegOrder[range >> 2];
  const LevelSet& L = set.L[octave];
  const float pixelWidth = set.pixelWidth[octave];
  const int t = threadIdx.x, sub = t & (LANES - 1), col = t >> LPKS;
  const float fsub2 = (float)(2 * sub);
  const uint32_t local = (blockIdx.x - tab->start[range]) * PER_BLOCK + (uint32_t)(t >> LPKS);
  const bool have = local < tab->count[range];
  const int gi = (int)(tab->first[range] + local);
  const float pi = SSRLCV_PI_F;
  const float rad10 = pi / 18.0f;
  const float inv10 = 1.0f / rad10;
  ssrlcv_sskeypoint kp;
  kp.loc.x = kp.loc.y = 0.0f;
  kp.sigma = 1.0f;
  if (have) kp = set.kps[octave][gi];
  const float kx = kp.loc.x, ky = kp.loc.y;
  const float windowWidth = ceilf(kp.sigma * 3.0f * lambda / pixelWidth);
  const float minx = kx - windowWidth, miny = ky - windowWidth, maxx = kx + windowWidth, maxy = ky + windowWidth;
  // (key points live on DoG levels 1..3: refinement discards what would move to level 0 or 4)
  const bool inside = have && useg >= 1 && useg <= 3 &&
                      !(minx < 0.0f || miny < 0.0f || maxx >= (unsigned)(L.w - 1) || maxy >= (unsigned)(L.h - 1));
#pragma unroll
  for (int i = 0; i < 36 * PER_BLOCK / 64; ++i) (&s_hist[0][0])[i * 64 + t] = 0.0f;
  const float weight = 2.0f * lambda * lambda * kp.sigma * kp.sigma;
  const float rweight = 1.0f / weight;
  const int W = L.w;
  const size_t levelStride = svp::polar_level_stride(L.w, L.h);
  // every lane of the block is in the same blur segment (the ranges are cut there), so the polar table -- the buffer
  // descriptor -- is wave-uniform
  {
    // entry e of the table (flat index e - 1) is at byte 8 e; reads past the end return 0 (and belong to lanes whose
    // samples are not used: a window row stays inside the level)
    const float2* base = L.polar + (size_t)(useg - 1) * levelStride;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(uint32_t)(levelStride * 8), 0x00020000);  // < 4 GiB: checked by plan_create
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    // A cursor walks the window chunk by chunk in the reference's order (x += 1.0f CW times, then the next row); the
    // chunk being evaluated is one chunk behind the one being requested.  The cursor is the same in every lane of a group.
    struct Cursor { float x, y, ty2; unsigned rowoff; bool act; };
    Cursor cf;  // the next chunk to request
    cf.x = minx;
    cf.y = miny;
    cf.ty2 = (miny - ky) * (miny - ky);
    cf.rowoff = (unsigned)round_pos(miny) * (unsigned)W + 1u;
    cf.act = inside;
    // The reference's `x += 1.0f`, k times, is one correctly rounded x + k whenever the k steps cross at most one
    // binade boundary: the steps in front of the crossing are exact, the crossing rounds x + k* once to the coarser grid
    // (on which every later + 1.0f is exact again), and rounding to that grid commutes with adding the integer k - k*
    // (an even multiple of its spacing, which is at most 1/2 up to 2^23: ties break the same way).  From x >= CW a chunk
    // of CW steps cannot hold two powers of two.  Below (windows at the level's left edge) the chain is walked.
    auto advance = [&](Cursor& c) {  // same row while the reference's loop condition holds, else the next row
      float xn = c.x + (float)CW;
      if (__any(c.act && !(c.x >= (float)CW))) {
        float ch = c.x;
#pragma unroll 8
        for (int i = 0; i < CW; ++i) ch += 1.0f;
        xn = c.x >= (float)CW ? xn : ch;
      }
      if (c.act) {
        if (xn <= maxx) {
          c.x = xn;
        } else {
          c.x = minx;
          c.y += 1.0f;
          if (!(c.y <= maxy)) c.act = false;
          const float ty = c.y - ky;
          c.ty2 = ty * ty;
          c.rowoff = (unsigned)round_pos(c.y) * (unsigned)W + 1u;
        }
      }
    };
    // Load j of lane `sub` is the 16-byte piece j * LANES + sub of the chunk: the samples 2 (j LANES + sub) and the one
    // behind it.  The group's pieces of one instruction are neighbours in memory.
    typedef u32x4 Chunk[kThetaChunk / 2];
    auto fetch = [&](const Cursor& c, Chunk& into) {
      if (!c.act) return;
      const unsigned off = (c.rowoff + (unsigned)round_pos(c.x)) * 8u + 16u * (unsigned)sub;  // byte offset, below 2^32
#pragma unroll
      for (int j = 0; j < kThetaChunk / 4; ++j) into[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(off + 16u * LANES * j), 0, 0);
    };
    // the chunk's upper half (see `upperHalf` in the evaluation): requested only if some lane of the wave can use it.  One sample
    // of slack covers the difference between this sum and the walked coordinate the evaluation tests: what it uses has been loaded.
    auto fetch_upper = [&](const Cursor& c, Chunk& into) {
      if (!c.act) return;
      const unsigned off = (c.rowoff + (unsigned)round_pos(c.x)) * 8u + 16u * (unsigned)sub;
#pragma unroll
      for (int j = kThetaChunk / 4; j < kThetaChunk / 2; ++j) into[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(off + 16u * LANES * j), 0, 0);
    };
    auto wants_upper = [&](const Cursor& c) { return __any(c.act && c.x + (float)(CW / 2 - 1) <= maxx); };
    // One chunk is requested ahead of the one being evaluated.  The loop is written out for two chunk buffers that trade
    // places (the compiler's rotation of one pair cost 40 register moves per chunk); two chunks ahead was measured in
    // round 5 and is no faster.
    Chunk qa, qb;
    Cursor ca, cb;
#pragma unroll
    for (int j = 0; j < kThetaChunk / 2; ++j) qa[j] = qb[j] = u32x4{0u, 0u, 0u, 0u};
    ca = cf;
    fetch(cf, qa);
    if (wants_upper(cf)) fetch_upper(cf, qa);
    advance(cf);
    cb = cf;
    auto step = [&](Chunk& cur, const Cursor& cc, Chunk& nxt, Cursor& cn) __attribute__((always_inline)) {
      cn = cf;
      fetch(cf, nxt);
      if (wants_upper(cf)) fetch_upper(cf, nxt);
      advance(cf);
      // the chunk's coordinates (sample s of the chunk: x_0 `+= 1.0f` s times; see advance); this lane keeps those of its
      // eight samples s = 2 (j LANES + sub) + h
      float xs[kThetaChunk];
      float xlast = cc.x + (float)(CW - 1);
#pragma unroll
      for (int i = 0; i < kThetaChunk; ++i) xs[i] = cc.x + (fsub2 + (float)(((i >> 1) << (1 + LPKS)) | (i & 1)));
      if (__any(cc.act && !(cc.x >= (float)CW))) {
        float ch = cc.x;
        float xc[kThetaChunk];
#pragma unroll
        for (int sIdx = 0; sIdx < CW; ++sIdx) {
          const int q = (sIdx >> 1) & (LANES - 1), i = ((sIdx >> (1 + LPKS)) << 1) | (sIdx & 1);
          if (LANES == 1 || q == 0) xc[i] = ch;  // (first writer of xc[i])
          else xc[i] = sub == q ? ch : xc[i];
          if (sIdx + 1 < CW) ch += 1.0f;
        }
        const bool direct = cc.x >= (float)CW;
#pragma unroll
        for (int i = 0; i < kThetaChunk; ++i) xs[i] = direct ? xs[i] : xc[i];
        xlast = direct ? xlast : ch;
      }
      const float cty2 = cc.ty2;
      const unsigned crow = cc.rowoff;
      const bool cact = cc.act;
      // Entry s of the chunk is sample s's pixel iff llroundf(x_s) == llroundf(x_0) + s.  In exact arithmetic it is;
      // `+= 1.0f` rounds only when it crosses a binade, and then only towards a fraction that every coarser binade still
      // represents: a fraction below one half can be lifted onto it (the rounded coordinate then is one more than the
      // entry's, for the rest of the row), never back -- checking the chunk's last entry covers them all.
      const bool aligned = cc.x >= 4.0f && round_pos(xlast) == round_pos(cc.x) + (CW - 1);
      // rare (a chunk that crosses a binade with an unlucky fraction, or starts below x = 4): the group's samples are
      // gathered one by one (behind the request for the next chunk in program order: its wait is also that request's)
      if (__any(cact && !aligned)) {
        if (cact && !aligned) {
#pragma unroll
          for (int i = 0; i < kThetaChunk; ++i) {
            u32x2 e = u32x2{0u, 0u};
            if (xs[i] <= maxx) e = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)((crow + (unsigned)round_pos(xs[i])) * 8u), 0, 0);
            cur[i >> 1][(i & 1) * 2] = e[0];
            cur[i >> 1][(i & 1) * 2 + 1] = e[1];
          }
        }
      }
      // Two phases per chunk.  (1) The samples' Gaussian weights and bins are independent of each other and of the
      // histogram: evaluated together they overlap their long dependent chains (the Horner steps, the divisions).
      // (2) The histogram updates, strictly in raster order, branch-free within a lane: a sample that does not count
      // (past the row's end, or bin 36 from an angle one ulp below 2 pi, which the reference's array has no slot for)
      // adds fmaf(0, w, h) = h to bin 0.
      auto weigh = [&](float ang, float xi, float& wgt, int& bin) {
        const float angle = fmod_2pi_above(ang + (2.0f * pi), 2.0f * pi);
        bin = (int)floorf(sv::exact_div3(angle, rad10, inv10));
        const float tx = xi - kx;
        wgt = sv::expf_nonpos(sv::exact_div5(-((tx * tx) + cty2), weight, rweight));
      };
      // The second half of a lane's samples (loads 2 and 3: the chunk's samples from 4 LANES on) lies past the row's end in every
      // lane of the wave for the last chunk of many rows (a 45-sample row of a blur-2 window is a chunk of 32 and one of 13): then
      // neither their weights nor their votes -- all of them fmaf(0, w, h) = h -- are formed.  (xs[4] is the lane's first sample of
      // that half and the sub-0 lane's is the half's first.)
      const bool upperHalf = __any(cact && xs[kThetaChunk / 2] <= maxx);
      float wg[kThetaChunk], mg[kThetaChunk];
      float* slot[kThetaChunk];
      auto weigh_half = [&](auto hc) __attribute__((always_inline)) {
        constexpr int i0 = decltype(hc)::value * (kThetaChunk / 2);
#pragma unroll
        for (int i = i0; i < i0 + kThetaChunk / 2; ++i) {
          // (element first, cast second: __builtin_bit_cast applied directly to a vector-element lvalue reads element 0)
          const unsigned um = cur[i >> 1][(i & 1) * 2], ua = cur[i >> 1][(i & 1) * 2 + 1];
          int bin;
          weigh(__builtin_bit_cast(float, ua), xs[i], wg[i], bin);
          const bool counts = cact && xs[i] <= maxx && (unsigned)bin < 36u;
          slot[i] = &s_hist[counts ? bin : 0][col];
          mg[i] = counts ? __builtin_bit_cast(float, um) : 0.0f;  // (the weight is finite: a zero magnitude is enough)
        }
      };
      auto vote_half = [&](auto hc) __attribute__((always_inline)) {
        constexpr int j0 = decltype(hc)::value * (kThetaChunk / 4);
#pragma unroll
        for (int j = j0; j < j0 + kThetaChunk / 4; ++j) {
#pragma unroll
          for (int q = 0; q < LANES; ++q) {
            if (LANES == 1 || sub == q) {
              // two votes per LDS round trip: both slots are read first, the second takes the first's sum when it is the same slot
              const float h0 = *slot[2 * j], h1 = *slot[2 * j + 1];
              const float n0 = fmaf(mg[2 * j], wg[2 * j], h0);
              const float n1 = fmaf(mg[2 * j + 1], wg[2 * j + 1], slot[2 * j + 1] == slot[2 * j] ? n0 : h1);
              *slot[2 * j] = n0;
              *slot[2 * j + 1] = n1;  // (LDS keeps a wave's stores in order: the same slot ends up holding n1)
            }
            // The order of the phases is an order between LANES, which the language knows nothing about: to the compiler
            // `if (sub == 0) X; if (sub == 1) X;` is X executed once by every thread, and it has merged the phases on that
            // ground (two lanes, round 6: every lane voted at once).  A wave barrier -- no instruction, but convergent and
            // with side effects -- keeps each phase a region of its own, in program order.
            if (LANES > 1) __builtin_amdgcn_wave_barrier();
          }
        }
      };
      weigh_half(std::integral_constant<int, 0>{});
      if (upperHalf) weigh_half(std::integral_constant<int, 1>{});
      vote_half(std::integral_constant<int, 0>{});
      if (upperHalf) vote_half(std::integral_constant<int, 1>{});
    };
    while (__any(ca.act)) {
      step(qa, ca, qb, cb);
      if (!__any(cb.act)) break;
      step(qb, cb, qa, ca);
    }
  }
  uint32_t cnt = 0;
  float outTheta[MAXO];
#pragma unroll
  for (int i = 0; i < MAXO; ++i) outTheta[i] = -FLT_MAX;
  if (inside && sub == 0) {
    float maxHist = 0.0f;
    for (int i = 0; i < 36; ++i) {
      const float h = s_hist[i][col];
      if (h > maxHist) maxHist = h;
    }
    maxHist *= orientationThreshold;
    float bestMag[MAXO], bestTh[MAXO];
#pragma unroll
    for (int i = 0; i < MAXO; ++i) { bestMag[i] = 0.0f; bestTh[i] = 0.0f; }
    float hprev = s_hist[35][col], hb = s_hist[0][col];
    for (int b = 0; b < 36; ++b) {
      const float hnext = s_hist[b == 35 ? 0 : b + 1][col];
      // tests of :1064-1068 (circular neighbours) and the weakest kept peak
      if (!(hb < maxHist || hb < hprev || hb < hnext || hb < bestMag[MAXO - 1])) {
        float tmag = hb;
        float tt = (hprev - hnext) / (hprev - (2.0f * hb) + hnext);
        tt *= (pi / 36.0f);
        tt += (b * rad10);
        tt = fmodf(tt + (2.0f * pi), 2.0f * pi);
#pragma unroll
        for (int i = 0; i < MAXO; ++i) {
          if (tmag > bestMag[i]) {
#pragma unroll
            for (int ii = i; ii < MAXO; ++ii) {
              float m2 = bestMag[ii], t2 = bestTh[ii];
              bestMag[ii] = tmag;
              bestTh[ii] = tt;
              tmag = m2;
              tt = t2;
            }
          }
        }
      }
      hprev = hb;
      hb = hnext;
    }
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
      if (bestMag[i] != 0.0f) { outTheta[i] = bestTh[i]; cnt = i + 1; }
    }
  }
  if (have && sub == 0) {
#pragma unroll
    for (int i = 0; i < MAXO; ++i) set.thetas[octave][(size_t)gi * svp::kMaxOrient + i] = outTheta[i];
    set.thetaCnt[octave][gi] = cnt;
  }
}

// ---- S14: fillDescriptors(SSKeyPoint) (src/SIFT_FeatureFactory.cu:475-549) ------------------------------------------------
// One WAVE per key point (the reference: a 4x4x8 block of which 16 threads sweep the window).  The samples of the
// rotated (2w+1)^2 window vote into the wave's 4x4x8 LDS histogram (the reference also uses shared-memory atomics,
// :521).  Every vote is the reference's expression, operand for operand and rounding for rounding:
//     temp = (1 - hx/binWidth) * (1 - hy/binWidth) * (1 - angle/rad45) * (|grad| * expf(-(r^2) / (2 w^2)))
// with IEEE divisions and the shared expf / atan2f / sincos of sv_math.h.  What the reference leaves undefined is the
// ORDER of its float atomicAdds; here (and in the oracle's sum mode 0, oracle/oracle_sift.c) a vote enters its bin as
// the integer nearest to temp * 2^k (k per key point, see kDescCopies) and the integer sum is exact, so the result does
// not depend on the order -- descriptors are deterministic and bit-identical to the oracle's.
// The kernel sits between VALU issue and the LDS atomic rate, so the loop is written for instruction count:
//   - a lane carries the four-fold ORBIT of a window sample (see the loop): one coordinate set, one Gaussian and one
//     pass over the 16 cells serve four samples;
//   - everything that depends on the key point only comes from a lane-parallel pre-pass (k_desc_consts) or is
//     wave-uniform in SGPRs (rotated cell centres), so a cell test is {2 v_sub, v_max, v_cmp} on SGPR operands;
//   - the 16 cells are visited in a uniform, fully unrolled loop: a cell's vote code runs once under the lane mask of
//     its passing orbits instead of a per-lane loop over set bits, the coordinate differences of the test are reused
//     for the weights and the LDS address is lane base + immediate.
__device__ __forceinline__ float uniform_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
// Per-key-point constants of k_descriptors, computed one key point per LANE by a pre-pass: in the wave-per-key-point
// kernel the same arithmetic (sine and cosine alone are ~100 instructions) ran once per WAVE, 64 lanes wide on a single
// value.  The descriptor kernel fetches the 32-byte record with one scalar load.
struct DescConst {
  float c, s;           // cos / sin of -theta (:497-498)
  float windowWidth;    // ceil(sigma * lambda / pixelWidth) (:487)
  float invExpDen;      // 1 / (2 windowWidth^2), correctly rounded: the Gaussian is expf(-(r^2) / (2 w^2)) (:508)
  float voteScale;      // 2^k: fixed-point scale of the votes (see kDescCopies)
  float invBin;         // 1 / binWidth, correctly rounded (binWidth = windowWidth / 2)
  uint32_t magic;       // ceil(2^32 / windowWidth): orbit index -> quadrant row by a multiply-high
  int32_t segment;      // blur segment of the key point = DoG level its window is sampled from
};
static_assert(sizeof(DescConst) == 32, "one s_load_dwordx8");
// One launch over a descriptor work list (every octave, or one sampling group): unit u of the table = one key point.
// segment: the blur segment the octave's state gives the key point (segment_of) -- or, for a group of octave 0 whose
// expansion is still running in pieces (regularCtl set and ->regular: the state is rewritten by the last piece), the segment
// of the unit's range, which is the same number there (the blur indices are an ordered partition, see GroupCtl).
__global__ __launch_bounds__(256) void k_desc_consts(const RangeTable* tab, OctaveSet set, float lambda, const OctaveState* states,
                                                     const GroupCtl* regularCtl) {
  const uint32_t total = tab->total;
  const bool fromTable = regularCtl != nullptr && regularCtl->regular != 0u;
  for (uint32_t u = blockIdx.x * 256 + threadIdx.x; u < total; u += gridDim.x * 256) {
    const int range = range_of(tab, u);
    const int octave = range & 3;
    const int i = (int)(tab->first[range] + (u - tab->start[range]));
    const ssrlcv_sskeypoint* kps = set.kps[octave];
    const float pixelWidth = set.pixelWidth[octave];
    DescConst* __restrict__ out = (DescConst*)const_cast<void*>(set.consts[octave]);
    DescConst d;
    const float theta = kps[i].theta;
    d.c = sv_cosf(-theta);
    d.s = sv_sinf(-theta);
    d.windowWidth = ceilf(kps[i].sigma * lambda / pixelWidth);
    d.invExpDen = 1.0f / (2.0f * d.windowWidth * d.windowWidth);
    // sqrt(2) * (windowWidth + 2)^2 * 2^k < 2^31
    int boundExp;
    (void)frexpf(1.4143f * ((d.windowWidth + 2.0f) * (d.windowWidth + 2.0f)), &boundExp);
    d.voteScale = ldexpf(1.0f, 31 - boundExp);
    d.invBin = 1.0f / (d.windowWidth / 2.0f);
    const unsigned wi = (unsigned)(int)d.windowWidth;
    d.magic = wi > 1u ? 0xFFFFFFFFu / wi + 1u : 0u;  // ceil(2^32 / w) (2^32 / w when w is a power of two); w = 1 is special-cased
    d.segment = fromTable ? kSegOrder[range >> 2] : segment_of(states + octave, i);
    out[i] = d;
  }
}

// A vote enters its bin as the integer nearest to temp * 2^k, halves up: v_cvt_rpi_i32_f32 is floor(x + 0.5) evaluated
// exactly (checked against floorf(x) + (x - floorf(x) >= 0.5f) for every float in [0, 2^31): tools/f64_rate.hip); the
// oracle spells that out.  Plain truncation, which the first version used, biases every bin low by half a unit per
// vote -- hundreds of units on bins that hold 1e5..1e6 of them, enough to move descriptor bytes and to lose 2 of the
// reference's 13 534 golden matches.
__device__ __forceinline__ unsigned vote_u32(float v) {
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(v));
  return (unsigned)r;
}

// LDS atomics are processed one 16-lane row at a time and lanes of a row that hit the same address serialise
// (tools/lds_atomic_rate.hip, cycles per wave-instruction per SIMD: ds_add_u32 16.5 / ds_add_u64 24.5 conflict-free and
// with duplicates only across rows; two lanes of a row per address 24 / 49; with 4 copies and 64-bit bins, the layout
// this kernel had first, 49-57).  So the bins are 32-bit, in 8 lane-private copies (copy = lane & 7: at most two lanes
// of a row share a copy, they collide only when their orientation bins coincide): 4 KiB per wave and <= 64 VGPRs, 8
// waves per SIMD.  32-bit sums need a per-key-point fixed-point scale 2^k: a bin receives at most (2 binWidth + 2)^2
// votes (lattice points of a rotated square) of at most sqrt(2), k is the largest power with bound * 2^k < 2^31
// (k = 22 at w = 12, 20 at w = 29).  The normalisation that follows is scale invariant, so the sums are used as they are.
#ifndef SSRLCV_DESC_COPIES
#define SSRLCV_DESC_COPIES 8
#endif
constexpr int kDescCopies = SSRLCV_DESC_COPIES;
#ifndef SSRLCV_DESC_VGPRS
#define SSRLCV_DESC_VGPRS 64
#endif
// waves_per_eu(4, 8) + an explicit VGPR cap instead of __launch_bounds__(256, 8): the latter also caps the SGPRs at 80
// (the budget of a 10-wave gfx9 part), and this kernel keeps the rotated cell centres and per-key-point constants there
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8), amdgpu_num_vgpr(SSRLCV_DESC_VGPRS), amdgpu_num_sgpr(102))) void k_descriptors(const RangeTable* tab, OctaveSet set, const uint32_t* featBase,
                                                     ssrlcv_sift_feature* __restrict__ features, uint32_t maxFeatures) {
  // lane-private copies of the 128 bins (copy = lane & 7, bin-major / copy-minor) keep same-address conflicts low
  __shared__ unsigned s_bins[4][(1 + 128 + 2) * kDescCopies];  // one bin of padding in front, two behind (votes of 0)
  __shared__ __attribute__((aligned(8))) uint8_t s_bytes[4][128];
  // the wave index as a scalar: key-point index, list loads and per-key-point constants then live in SGPRs
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned* bins = s_bins[wave] + kDescCopies;
  const int copy = lane & (kDescCopies - 1);
  const float pi = SSRLCV_PI_F;
  const float rad45 = pi / 4.0f;
  const float inv45 = 1.0f / rad45;
  const uint32_t totalUnits = tab->total;
  // one work unit = one key point; units walk the octaves' blur segments from the largest windows to the smallest
  // (Round 3 re-tried an XCD-contiguous unit order now that the kernel is gather-bound -- the blocks of residue class y
  // mod 8 of a range take the y-th eighth of its key points, so that an XCD's L2 sees one band of the tables instead of the
  // hundreds of rows the ~8000 key points in flight span: 1.19-1.23 ms either way.)
  for (uint32_t unit = blockIdx.x * 4 + wave; unit < totalUnits; unit += gridDim.x * 4) {
    const int range = range_of(tab, unit);
    const int octave = range & 3;
    const int gi = (int)(tab->first[range] + (unit - tab->start[range]));
    const LevelSet& L = set.L[octave];
    const float pixelWidth = set.pixelWidth[octave];
    const float2* __restrict__ polar = L.polar;  // key points live on levels 1..3, whose polar tables are always built
    const DescConst dc = reinterpret_cast<const DescConst*>(set.consts[octave])[gi];  // gi is wave-uniform: scalar loads
    const ssrlcv_sskeypoint kp = set.kps[octave][gi];
    const int seg = dc.segment;
    const float kx = kp.loc.x, ky = kp.loc.y;
    const float theta = kp.theta;
    const float windowWidth = dc.windowWidth;
    const float binWidth = windowWidth / 2.0f;
    const float c = dc.c, s = dc.s;
    const float2* __restrict__ pl = polar + (size_t)(seg - 1) * svp::polar_level_stride(L.w, L.h) + 1;
#pragma unroll
    for (int i = 0; i < 2 * kDescCopies; ++i) bins[i * 64 + lane] = 0u;
    // rotated cell centres (:511-512), identical expressions to the reference's per-sample recomputation: lane i
    // evaluates cell i & 15, the 16 {x, y} pairs are then broadcast into SGPR pairs
    f32x2 rc[16];  // only the nine cells with nx, ny >= 1 are read (see the cell loop): 18 SGPRs
    {
      const int ci = lane & 15;
      const float hx = ((float)(ci >> 2) * 0.5f - 0.75f) * windowWidth, hy = ((float)(ci & 3) * 0.5f - 0.75f) * windowWidth;
      const float rx = (hx * c) + (hy * s), ry = (-hx * s) + (hy * c);
#pragma unroll
      for (int cell = 0; cell < 16; ++cell) {
        if ((cell >> 2) == 0 || (cell & 3) == 0) continue;
        rc[cell].x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rx), cell));
        rc[cell].y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ry), cell));
      }
    }
    // the reference's divisions by binWidth, 2 w^2 and rad45 as IEEE quotients from hoisted reciprocals (sv::exact_div3,
    // checked exhaustively for every window width up to 255; the plan refuses wider windows: their orbit indices would
    // not fit the 16-bit multiply-high below either)
    const float expDen = 2.0f * windowWidth * windowWidth, invExpDen = dc.invExpDen, invBin = dc.invBin;
    const float voteScale = dc.voteScale;
    // Votes are exact integers, so the visiting order of the window samples does not matter.  The polar gathers of the
    // next batch are issued before the current one is used.
    //
    // Four-fold symmetry: a quarter turn maps the integer window onto itself, p = (x, y) -> (-y, x), and takes the
    // rotated coordinates (cx, cy) to (-cy, cx) EXACTLY (the same two products, summed in the other order or negated);
    // the 4 x 4 cell centres turn with it, rc[(3 - ny)*4 + nx] == (-rc[nx*4 + ny].y, rc[nx*4 + ny].x) bit for bit (hx, hy
    // run over -0.75, -0.25, 0.25, 0.75 times windowWidth).  So the cell test of the turned sample against the turned
    // cell is the test of p against the cell with |dx| and |dy| exchanged: the same pass / fail, the same product of the
    // two 1 - t/binWidth factors (a float product commutes), the same Gaussian of r^2 (so does a sum).  A lane therefore
    // carries the ORBIT {p, turned once, twice, three times}: one set of coordinates, one exponential, one pass over
    // the 16 cells; only the gathers and the orientation weights are per sample.  Orbits are numbered over the quadrant
    // x = 1..w, y = 0..w (index = y*w + x - 1); one more index stands for the window centre, which is its own orbit.
    const int Wi = (int)windowWidth;
    const unsigned centreIdx = (unsigned)(Wi * (Wi + 1));
    const unsigned magic = dc.magic;  // ceil(2^32 / w)
    auto sample = [&](unsigned idx, float& cx, float& cy, bool& ok) {
      const unsigned yq = Wi == 1 ? idx : __umulhi(idx, magic);  // idx / w, exact for idx < 2^16
      const unsigned xq = idx - yq * (unsigned)Wi + 1u;
      const bool centre = idx >= centreIdx;
      const float x = centre ? 0.0f : (float)(int)xq;  // integers: exact, the values of the reference's repeated += 1.0f
      const float y = centre ? 0.0f : (float)(int)yq;
      cx = (x * c) + (y * s);
      cy = (-x * s) + (y * c);
      ok = idx <= centreIdx && fmaxf(fabsf(cx), fabsf(cy)) <= windowWidth;  // :505, one compare for the whole orbit
    };
    // llroundf of the reference and its flat index into the gradient array (:507).  checkKeyPoints keeps the window
    // sigma * lambda / pixelWidth inside the level, but the loop runs over its ceiling, so a coordinate may reach -1,
    // W or H: the padded table returns what the reference's array returns (the neighbouring row) or zero (outside).
    // A turned sample's coordinate is e.g. (-cy) + kx = kx - cy, the same rounded sum.
    auto gather = [&](float px, float py) { return polar_px(pl, L.w, round_coord(px), round_coord(py)); };
    // orientation bins: every k in 0..7 with |ang - k*rad45| < rad45 votes 1 - |ang - k*rad45| / rad45 (:515-518);
    // ang lies in (-pi, 2 pi) (fmodf keeps the sign) and there is no wrap-around.  With k0 = floor(ang / rad45) these
    // are bins k0 and k0 + 1, each if it exists -- unless ang lies within rounding of a multiple of rad45, where the
    // reference's float tests can admit a third bin or reject one of the two.  `odd` marks those samples (the estimate
    // t45 is within 1e-4 of an integer; its own error is below 1e-6): they take no part in the fast path and are
    // replayed literally afterwards.  The two weights are the reference's expressions: |ang - (float)k * rad45|, an
    // IEEE division by rad45, 1 - quotient.  A missing vote adds 0 to a valid word (a neighbouring cell's bin or the
    // padding in front of / behind the histogram) instead of branching.
    auto split = [&](float2 pg, bool live, float& w0, float& w1, unsigned*& pa, float& ang, bool& odd) {
      ang = fmod_2pi_above(pg.y - theta + (2.0f * pi), 2.0f * pi);
      const float t45 = ang * inv45;
      const float fk = floorf(t45);
      const float u = t45 - fk;  // exact (Sterbenz-like: fk <= t45 < fk + 1)
      odd = live && !(u > 1.0e-4f && u < 0.9999f);
      int k0 = (int)fk;
      const float a0 = fabsf(ang - (fk * rad45)), a1 = fabsf(ang - ((fk + 1.0f) * rad45));
      w0 = (live && !odd && (unsigned)k0 < 8u) ? 1.0f - sv::exact_div3(a0, rad45, inv45) : 0.0f;
      w1 = (live && !odd && (unsigned)(k0 + 1) < 8u) ? 1.0f - sv::exact_div3(a1, rad45, inv45) : 0.0f;
      k0 = k0 < -1 ? -1 : k0;  // k0 in -4..8 -> -1..8
      pa = bins + k0 * kDescCopies + copy;  // bin k0 of cell 0; bin k0 + 1 is kDescCopies words further
    };
    float ncx, ncy;
    bool nok;
    sample((unsigned)lane, ncx, ncy, nok);
    float2 npg[4] = {make_float2(0.0f, 0.0f), make_float2(0.0f, 0.0f), make_float2(0.0f, 0.0f), make_float2(0.0f, 0.0f)};
    auto gather4 = [&]() {
      npg[0] = gather(ncx + kx, ncy + ky);
      npg[1] = gather(kx - ncy, ncx + ky);
      npg[2] = gather(kx - ncx, ky - ncy);
      npg[3] = gather(ncy + kx, ky - ncx);
    };
    if (nok) gather4();
    for (unsigned base = 0; base <= centreIdx; base += 64) {
      const float cx = ncx, cy = ncy;
      const bool ok = nok;
      const float2 pg0 = npg[0], pg1 = npg[1], pg2 = npg[2], pg3 = npg[3];
      const bool turned = ok && base + (unsigned)lane < centreIdx;  // the centre sample is its own orbit
      sample(base + 64 + (unsigned)lane, ncx, ncy, nok);
      if (nok) gather4();
      // gaussian weight (shared by the orbit), the fixed-point scale folded in (a power of two: exact)
      const float r2n = -((cx * cx) + (cy * cy));
      const float g = sv::expf_nonpos(sv::exact_div3(r2n, expDen, invExpDen)) * voteScale;
      float a0, a1, b0, b1, c0, c1, d0, d1, angA, angB, angC, angD;
      bool oddA, oddB, oddC, oddD;
      unsigned *pa, *pb, *pc, *pd;
      split(pg0, ok, a0, a1, pa, angA, oddA);
      split(pg1, turned, b0, b1, pb, angB, oddB);
      split(pg2, turned, c0, c1, pc, angC, oddC);
      split(pg3, turned, d0, d1, pd, angD, oddD);
      const float magA = pg0.x * g, magB = pg1.x * g, magC = pg2.x * g, magD = pg3.x * g;  // :508, times 2^k
      // cells whose rotated centre lies within binWidth of the sample on both axes (:513-514).  `ok` is folded into
      // the lane's threshold and the two axis tests into one compare of max(|tx|, |ty|), so a cell costs {2 v_sub,
      // v_max, v_cmp} + {s_and_saveexec, s_cbranch_execz, s_or}.  p votes into cell (nx, ny), its quarter turns with
      // the same (1 - hx)(1 - hy) into (3 - ny, nx), (3 - nx, 3 - ny) and (ny, 3 - nx).
      const float bwl = ok ? binWidth : -1.0f;
#pragma unroll
      for (int cell = 0; cell < 16; ++cell) {
        const int nx = cell >> 2, ny = cell & 3;
        // The orbit's first member lies in the quadrant x >= 1, y >= 0 (or is the centre).  A cell passes only if its
        // centre is within sqrt(2) binWidth of the sample (the test is |R (h - p)|_inf <= binWidth), and the centres of
        // column nx = 0 / row ny = 0 sit at -1.5 binWidth: at least 1.5 binWidth from every such sample.  Those seven
        // cells can only be reached by the turned members, through the symmetry -- nine tests per orbit, not sixteen.
        if (nx == 0 || ny == 0) continue;
        const int cellB = (3 - ny) * 4 + nx, cellC = 15 - cell, cellD = ny * 4 + (3 - nx);
        const float tx = fabsf(rc[cell].x - cx), ty = fabsf(rc[cell].y - cy);
        if (fmaxf(tx, ty) <= bwl) {
          const float wxy = (1.0f - sv::exact_div3(tx, binWidth, invBin)) * (1.0f - sv::exact_div3(ty, binWidth, invBin));
          atomicAdd(pa + cell * 8 * kDescCopies, vote_u32((wxy * a0) * magA));
          atomicAdd(pa + cell * 8 * kDescCopies + kDescCopies, vote_u32((wxy * a1) * magA));
          atomicAdd(pb + cellB * 8 * kDescCopies, vote_u32((wxy * b0) * magB));
          atomicAdd(pb + cellB * 8 * kDescCopies + kDescCopies, vote_u32((wxy * b1) * magB));
          atomicAdd(pc + cellC * 8 * kDescCopies, vote_u32((wxy * c0) * magC));
          atomicAdd(pc + cellC * 8 * kDescCopies + kDescCopies, vote_u32((wxy * c1) * magC));
          atomicAdd(pd + cellD * 8 * kDescCopies, vote_u32((wxy * d0) * magD));
          atomicAdd(pd + cellD * 8 * kDescCopies + kDescCopies, vote_u32((wxy * d1) * magD));
        }
      }
      // the literal replay (:509-524) of the samples whose direction sits on a bin boundary: about one in 10^4
      if (__ballot(oddA || oddB || oddC || oddD) != 0ull) {
#pragma unroll 1
        for (int m = 0; m < 4; ++m) {
          const bool odd = m == 0 ? oddA : m == 1 ? oddB : m == 2 ? oddC : oddD;
          if (!odd) continue;
          const float mcx = m == 0 ? cx : m == 1 ? -cy : m == 2 ? -cx : cy;
          const float mcy = m == 0 ? cy : m == 1 ? cx : m == 2 ? -cy : -cx;
          const float mang = m == 0 ? angA : m == 1 ? angB : m == 2 ? angC : angD;
          const float mmag = m == 0 ? magA : m == 1 ? magB : m == 2 ? magC : magD;
#pragma unroll 1
          for (int cell = 0; cell < 16; ++cell) {
            const int ci = lane & 15;
            (void)ci;
            const float hx0 = ((float)(cell >> 2) * 0.5f - 0.75f) * windowWidth, hy0 = ((float)(cell & 3) * 0.5f - 0.75f) * windowWidth;
            const float rx = (hx0 * c) + (hy0 * s), ry = (-hx0 * s) + (hy0 * c);
            float hx = fabsf(rx - mcx), hy = fabsf(ry - mcy);
            if (hx <= binWidth && hy <= binWidth) {
              hx = hx / binWidth;
              hy = hy / binWidth;
              for (int k = 0; k < 8; ++k) {
                float angle = fabsf(mang - ((float)k * rad45));
                if (angle < rad45) {
                  angle /= rad45;
                  const float temp = (1.0f - hx) * (1.0f - hy) * (1.0f - angle) * mmag;
                  atomicAdd(bins + (cell * 8 + k) * kDescCopies + copy, vote_u32(temp));
                }
              }
            }
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // normalise, clamp at 0.2, renormalise, quantise (:529-542); each lane owns bins lane and lane + 64.  The two norms
    // are balanced-tree sums (the reference's are float atomicAdds in no defined order): pairs 64 apart, then the
    // xor butterfly 32, 16, .. 1 -- the tree the oracle's sum mode 0 spells out.
    unsigned t0 = 0u, t1 = 0u;
#pragma unroll
    for (int cpy = 0; cpy < kDescCopies; ++cpy) {
      t0 += bins[lane * kDescCopies + cpy];
      t1 += bins[(lane + 64) * kDescCopies + cpy];
    }
    float v0 = (float)t0, v1 = (float)t1;
    float sq = sqrtf(sv::wave_sum((v0 * v0) + (v1 * v1)));
    v0 = v0 / sq;
    v1 = v1 / sq;
    if (v0 > 0.2f) v0 = 0.2f;
    if (v1 > 0.2f) v1 = 0.2f;
    sq = sqrtf(sv::wave_sum((v0 * v0) + (v1 * v1)));
    // bins are [nx][ny][k]; the descriptor byte order is (ny*4 + nx)*8 + k (:542)
    {
      int b0 = lane, b1 = lane + 64;
      int o0 = (((b0 >> 3) & 3) * 4 + (b0 >> 5)) * 8 + (b0 & 7);
      int o1 = (((b1 >> 3) & 3) * 4 + (b1 >> 5)) * 8 + (b1 & 7);
      s_bytes[wave][o0] = (uint8_t)roundf(255.0f * v0 / sq);
      s_bytes[wave][o1] = (uint8_t)roundf(255.0f * v1 / sq);
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t fi = featBase[octave] + (uint32_t)gi;
    if (fi < maxFeatures) {
      ssrlcv_sift_feature* f = features + fi;
      if (lane < 16) *reinterpret_cast<uint2*>(&f->values[lane * 8]) = *reinterpret_cast<const uint2*>(&s_bytes[wave][lane * 8]);
      if (lane == 16) {
        f->parent = -1;  // Feature() default (include/Feature.cuh:43-46); the reference kernel never writes it
        f->theta = kp.theta;
        f->sigma = kp.sigma;
        f->loc.x = kp.loc.x * pixelWidth;
        f->loc.y = kp.loc.y * pixelWidth;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

LevelSet make_levels(const ssrlcv_sift_plan* plan, char* ws, int o) {
  LevelSet L;
  const svp::OctavePlan& oc = plan->oct[o];
  for (int b = 0; b < svp::kGauss; ++b) L.lvl[b] = (const float*)(ws + plan->off_gauss[o][b]);
  L.lvlMinMax = (const float*)(ws + plan->off_minmax) + (size_t)o * 2 * (svp::kGauss + svp::kDog);
  L.minmax = L.lvlMinMax + 2 * svp::kGauss;
  L.polar = (const float2*)(ws + oc.off_polar);
  L.w = (int)oc.w;
  L.h = (int)oc.h;
  return L;
}

// the polar tables of the octaves oFirst..oLast in one launch (the three small ones were each a launch of a few waves per CU):
// the octaves in front of the range own no block (start 0, like the first one of the range), those behind it start past the grid
void launch_polar(const ssrlcv_sift_plan* plan, char* ws, hipStream_t st, int oFirst = 0, int oLast = svp::kOctaves - 1) {
  PolarJobs jobs;
  uint32_t pos = 0;
  for (int o = 0; o < svp::kOctaves; ++o) {
    const svp::OctavePlan& oc = plan->oct[o];
    jobs.L[o] = make_levels(plan, ws, o);
    jobs.out[o] = (float2*)(ws + oc.off_polar);
    jobs.start[o] = pos;
    jobs.tilesX[o] = (oc.w + 255) / 256;
    if (o >= oFirst && o <= oLast) pos += jobs.tilesX[o] * ((oc.h + kPolRows - 1) / kPolRows);
  }
  jobs.start[svp::kOctaves] = pos;
  hipLaunchKernelGGL(k_polar, dim3(pos), dim3(256), 0, st, jobs);
}

}  // namespace
namespace svp {
void launch_polar_octave(const ssrlcv_sift_plan* plan, char* ws, int octave, hipStream_t st) {
  PolarJobs jobs;
  const svp::OctavePlan& oc = plan->oct[octave];
  const uint32_t tilesX = (oc.w + 255) / 256, tiles = tilesX * ((oc.h + kPolRows - 1) / kPolRows);
  for (int o = 0; o < svp::kOctaves; ++o) {  // every block resolves to `octave`: start[k] = 0 up to it, past the grid behind it
    jobs.L[o] = make_levels(plan, ws, octave);
    jobs.out[o] = (float2*)(ws + oc.off_polar);
    jobs.tilesX[o] = tilesX;
    jobs.start[o] = o <= octave ? 0u : tiles;
  }
  jobs.start[svp::kOctaves] = tiles;
  hipLaunchKernelGGL(k_polar, dim3(tiles), dim3(256), 0, st, jobs);
}
}  // namespace svp
namespace {
// discardExtrema: stable per-segment compaction from `src` into `dst`
hipError_t run_discard(OctaveState* st, const ssrlcv_sskeypoint* src, ssrlcv_sskeypoint* dst, uint32_t cap,
                       uint32_t* words, hipStream_t stream) {
  const OctaveState* cst = st;
  auto keyfn = [=] __device__(uint32_t i) -> uint32_t {
    if (!cst->hasExtrema || (int)i >= cst->n) return 0u;
    if (src[i].discard) return 0u;
    return 1u << segment_of(cst, (int)i);
  };
  auto emit = [=] __device__(uint32_t i, int, uint32_t d) { dst[d] = src[i]; };
  uint32_t* totals = nullptr;
  auto post = [=] __device__(const uint32_t* tot) { book_discard(st, tot); };
  return svc::partition<svp::kDog, 8>(cap, keyfn, emit, words, &totals, stream, &st->n, 1u, post);
}

// Grid of the wave-per-key-point kernels (orientations, descriptors) in units of list_blocks(): their waves walk the
// list with a stride, and key points differ 4x in window size, so with just enough blocks to fill the chip once (x2,
// all resident from the start) the slowest waves set the time while most SIMDs idle; many more, shorter-lived blocks
// than wave slots let the dispatcher balance the tail.  Measured on a 4096^2 image (describe, ms): x1 9.20, x2 8.10,
// x4 7.82, x8 7.35, x16 7.37, x64 7.28.
constexpr unsigned kWaveKernelOversubscription = 16;
inline unsigned list_blocks(uint32_t cap) {
  unsigned b = (cap + 255) / 256;
  return b > 1024 ? 1024 : b;
}

}  // namespace

extern "C" {

int ssrlcv_sift_plan_keypoints(const ssrlcv_sift_plan* plan, void* workspace, int octave, ssrlcv_sskeypoint** list,
                               int** blurIndices_dev) {
  if (!plan || !workspace || octave < 0 || octave >= svp::kOctaves) return SSRLCV_ERR_INVALID_ARG;
  char* ws = (char*)workspace;
  // the list ping-pongs between the octave's two buffers once per compaction; the plan remembers where it is
  if (list) *list = (ssrlcv_sskeypoint*)(ws + (plan->listInB[octave] ? plan->oct[octave].off_kpB : plan->oct[octave].off_kpA));
  if (blurIndices_dev) *blurIndices_dev = (int*)(ws + plan->off_state + sizeof(OctaveState) * octave);
  return SSRLCV_OK;
}

int ssrlcv_sift_plan_overflow(const ssrlcv_sift_plan* plan, const void* workspace, uint32_t* octaveMask,
                              ssrlcv_stream_t stream) {
  if (!plan || !workspace || !octaveMask) return SSRLCV_ERR_INVALID_ARG;
  OctaveState st[svp::kOctaves];
  SSRLCV_HIP_TRY(hipMemcpyAsync(st, (const char*)workspace + plan->off_state, sizeof st, hipMemcpyDeviceToHost,
                                (hipStream_t)stream));
  SSRLCV_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  uint32_t m = 0;
  for (int o = 0; o < svp::kOctaves; ++o)
    if (st[o].overflow) m |= 1u << o;
  *octaveMask = m;
  return m ? SSRLCV_ERR_CAPACITY : SSRLCV_OK;
}

// ---- the key-point stages, one function per reference launch site --------------------------------------------------------
}  // extern "C"
namespace {

struct ListCtx {  // one octave's list chain
  const ssrlcv_sift_plan* plan;
  const svp::OctavePlan* oc;
  int o;
  hipStream_t s;
  OctaveState* st;
  LevelSet L;
  ssrlcv_sskeypoint *A, *B;
  uint8_t* flags;
  uint32_t* words;
  uint32_t cap;
  ssrlcv_sskeypoint* cur() const { return plan->listInB[o] ? B : A; }
  ssrlcv_sskeypoint* oth() const { return plan->listInB[o] ? A : B; }
  void flip() const { plan->listInB[o] ^= 1; }
};
ListCtx make_ctx(const ssrlcv_sift_plan* plan, char* ws, int o, hipStream_t s) {
  ListCtx c;
  c.plan = plan;
  c.oc = &plan->oct[o];
  c.o = o;
  c.s = s;
  c.st = (OctaveState*)(ws + plan->off_state) + o;
  c.L = make_levels(plan, ws, o);
  c.A = (ssrlcv_sskeypoint*)(ws + c.oc->off_kpA);
  c.B = (ssrlcv_sskeypoint*)(ws + c.oc->off_kpB);
  c.flags = (uint8_t*)(ws + c.oc->off_flags);
  c.words = (uint32_t*)(ws + c.oc->off_part);
  c.cap = c.oc->cap;
  return c;
}

// searchForExtrema (src/FeatureFactory.cu:86-159): fillExtrema + the compaction of what findExtrema flagged.
// findExtrema itself ran inside ssrlcv_hip_sift_build_dog (k_dogx: the DoG values exist only there); what is left is
// the compaction of the flag bytes into the list.  removeNoise(noiseThreshold * 0.8) (src/FeatureFactory.cu:484) follows
// the search directly and tests the raw DoG value fillExtrema stores as the intensity, so the flag byte carries a second
// set of bits for the extrema that pass it (afterFirstNoise): the survivors, their order and extremaBlurIndices are
// those of search + discard, without the first list (1.33 M entries per 4096^2 image, 0.1 % of them noise) being written,
// flagged and compacted again.
int stage_extrema(const ListCtx& c, bool afterFirstNoise) {
  hipLaunchKernelGGL(k_state_reset, dim3(1), dim3(1), 0, c.s, c.st);
  c.plan->listInB[c.o] = 0;
  const uint32_t P = c.oc->w * c.oc->h, cap = c.cap;
  const int W = (int)c.oc->w;
  const int octaveId = c.o;
  const float s1 = c.oc->sigma[1], s2 = c.oc->sigma[2], s3 = c.oc->sigma[3];
  const LevelSet Lc = c.L;
  ssrlcv_sskeypoint* first = c.A;
  auto emit = [=] __device__(uint32_t p, int b, uint32_t d) {
    if (d >= cap) return;
    ssrlcv_sskeypoint kp;  // fillExtrema (src/FeatureFactory.cu:883-890)
    kp.octave = octaveId;
    kp.blur = (int)b + 1;
    kp.loc.x = (float)(p % W);
    kp.loc.y = (float)(p / W);
    kp.intensity = dog_view(Lc, b + 1).raw(p);
    kp.sigma = b == 0 ? s1 : b == 1 ? s2 : s3;
    kp.theta = -1.0f;
    kp.discard = 0;
    first[d] = kp;
  };
  uint32_t* totals = nullptr;
  hipError_t e = svc::partition_flags<3>(P, c.flags, afterFirstNoise ? svp::kNoiseFlagShift : 0, emit, c.words, &totals, c.s);
  if (e != hipSuccess) return (int)e;
  // (a kernel of its own here: the pixel-domain scatter has thousands of blocks, and counting them down with one
  // same-address atomic each costs more than this launch)
  hipLaunchKernelGGL(k_book_extrema, dim3(1), dim3(1), 0, c.s, c.st, totals, cap);
  return SSRLCV_OK;
}
int discard_flagged(const ListCtx& c) {  // discardExtrema (src/FeatureFactory.cu:161-215)
  hipError_t e = run_discard(c.st, c.cur(), c.oth(), c.cap, c.words, c.s);
  if (e != hipSuccess) return (int)e;
  c.flip();
  return SSRLCV_OK;
}
int stage_noise(const ListCtx& c, float threshold) {  // removeNoise (:267-285)
  hipLaunchKernelGGL(k_flag_noise, dim3(list_blocks(c.cap)), dim3(256), 0, c.s, c.st, c.cur(), threshold);
  return discard_flagged(c);
}
int stage_refine(const ListCtx& c) {  // refineExtremaLocation (:217-265)
  hipLaunchKernelGGL(k_refine, dim3(list_blocks(c.cap)), dim3(256), 0, c.s, c.st, c.cur(), c.L, c.oc->sigma[0],
                     c.oc->sigma[1] / c.oc->sigma[0]);
  int rc = discard_flagged(c);
  if (rc) return rc;
  // thrust::stable_sort by blur == stable partition on the blur value
  OctaveState* st = c.st;
  const OctaveState* cst = c.st;
  const ssrlcv_sskeypoint* src = c.cur();
  ssrlcv_sskeypoint* dst = c.oth();
  auto keyfn = [=] __device__(uint32_t i) -> uint32_t {
    if (!cst->hasExtrema || (int)i >= cst->n) return 0u;
    int b = src[i].blur;
    return (b < 0 || b >= svp::kDog) ? 0u : (1u << b);
  };
  auto emit = [=] __device__(uint32_t i, int, uint32_t d) { dst[d] = src[i]; };
  uint32_t* totals = nullptr;
  auto post = [=] __device__(const uint32_t* tot) { book_rescan(st, tot); };
  hipError_t e = svc::partition<svp::kDog, 8>(c.cap, keyfn, emit, c.words, &totals, c.s, &st->n, 1u, post);
  if (e != hipSuccess) return (int)e;
  c.flip();
  return SSRLCV_OK;
}
int stage_edges(const ListCtx& c) {  // removeEdges (:287-306)
  hipLaunchKernelGGL(k_flag_edges, dim3(list_blocks(c.cap)), dim3(256), 0, c.s, c.st, c.cur(), c.L, svp::kEdgeThreshold);
  return discard_flagged(c);
}
int stage_window(const ListCtx& c) {  // checkKeyPoints (src/SIFT_FeatureFactory.cu:81-110,449-461)
  hipLaunchKernelGGL(k_flag_window, dim3(list_blocks(c.cap)), dim3(256), 0, c.s, c.st, c.cur(), c.L.w, c.L.h, c.oc->pixelWidth,
                     c.plan->params.descriptorContribWidth);
  return discard_flagged(c);
}
int stage_noise_edges_window(const ListCtx& c) {  // the three of them with one discard (see k_flag_noise_edges_window)
  hipLaunchKernelGGL(k_flag_noise_edges_window, dim3(list_blocks(c.cap)), dim3(256), 0, c.s, c.st, c.cur(), c.L, svp::kNoiseThreshold,
                     svp::kEdgeThreshold, c.oc->pixelWidth, c.plan->params.descriptorContribWidth);
  return discard_flagged(c);
}

OctaveSet make_set(const ssrlcv_sift_plan* plan, char* ws) {
  OctaveSet set;
  for (int o = 0; o < svp::kOctaves; ++o) {
    const svp::OctavePlan& oc = plan->oct[o];
    set.L[o] = make_levels(plan, ws, o);
    set.kps[o] = (const ssrlcv_sskeypoint*)(ws + (plan->listInB[o] ? oc.off_kpB : oc.off_kpA));
    set.pixelWidth[o] = oc.pixelWidth;
    set.thetas[o] = (float*)(ws + oc.off_theta);
    set.thetaCnt[o] = (uint32_t*)(ws + oc.off_thetaCnt);
    set.consts[o] = ws + oc.off_descConst;
  }
  return set;
}

// ---- expandKeyPoints + the two thrust::remove calls in front of it (src/FeatureFactory.cu:594-611) in ONE pass (round 4).
// Element e = i * maxO + j is "orientation j of key point i"; it is kept when j < thetaCnt[i] and lands at its rank among
// the kept ones -- the list is sorted by blur segment and the order is kept, so the output is grouped by segment like
// upstream's per-blur loop builds it.  Rounds 1-3 ran the generic count -> scan -> scatter partition (three launches per
// octave between k_thetas and k_descriptors); here a tile takes its prefix by decoupled look-back (scan_lookback.h).
// The new extremaBlurIndices need no counting: idx'[s] = the number of kept elements in front of segment s = the running
// offset of the thread that meets element idx[s] * maxO.  The bookkeeping runs in the last block to leave.
// Elements per thread: a tile's prefix waits for the tile before it (decoupled look-back), ~0.15 us per link of that chain;
// with 4 elements per thread the 430 k elements of a 4096^2 image's octave 0 were 420 tiles = 85 us of mostly waiting
// (profiles/r05_timeline_*.txt), with 16 they are 105.
constexpr int kExpandItems = 16;
// Round 5: the launch covers PIECE g of the list -- the elements of the key points [lo, hi) with lo / hi = the octave's
// cut points (GroupCtl::cut; piece 0 starts at 0, the last piece ends at n) -- and continues the running offset where
// piece g - 1 stopped (ctl->totals[g], complete when this launch starts: the pieces are ordered by events).  One piece
// covering everything (pieces = 1) is the whole-octave expansion of round 4.  Only the LAST piece rewrites the octave's
// state, after every block of it has left, so every piece reads the same pre-expansion n and blur indices.
__global__ __launch_bounds__(svs::kThreads) void k_expand_orient(OctaveState* st, const ssrlcv_sskeypoint* __restrict__ src,
                                                                 ssrlcv_sskeypoint* __restrict__ dst, const float* __restrict__ thetas,
                                                                 const uint32_t* __restrict__ thetaCnt, uint32_t maxO, uint32_t cap,
                                                                 svs::TileScan<1> ts, GroupCtl* ctl, int g, int pieces) {
  constexpr int ITEMS = kExpandItems;
  constexpr uint32_t kTile = svs::kThreads * ITEMS;
  // read BEFORE anything of this launch can rewrite the state (the bookkeeping below waits for every block)
  const uint32_t n = st->hasExtrema && st->n > 0 ? (uint32_t)st->n : 0u;
  uint32_t lo = 0, hi = n;
  if (pieces > 1) {
    if (g > 0) lo = ctl->cut[g - 1] < n ? ctl->cut[g - 1] : n;
    if (g + 1 < pieces) hi = ctl->cut[g] < n ? ctl->cut[g] : n;
    if (hi < lo) hi = lo;
  }
  const uint32_t e0 = lo * maxO, e1 = hi * maxO;
  const uint32_t tiles = (e1 - e0 + kTile - 1) / kTile;  // <= ts.numTiles (sized for the capacity)
  const uint32_t before = ctl->totals[g];
  uint32_t segFirst[svp::kDog];
#pragma unroll
  for (int k = 0; k < svp::kDog; ++k) segFirst[k] = (uint32_t)st->idx[k] * maxO;
  for (uint32_t tile = svs::next_tile(ts.counter); tile < tiles; tile = svs::next_tile(ts.counter)) {
    const uint32_t base = e0 + tile * kTile + threadIdx.x * ITEMS;
    bool keep[ITEMS];
    uint32_t mine[1] = {0};
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const uint32_t e = base + j, i = e / maxO;
      keep[j] = e < e1 && (e - i * maxO) < thetaCnt[i];
      mine[0] += keep[j] ? 1u : 0u;
    }
    uint32_t excl[1], total[1], prefix[1];
    svs::block_exclusive<1>(mine, excl, total);
    svs::tile_prefix<1>(ts, tile, total, prefix);
    uint32_t at = before + prefix[0] + excl[0];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const uint32_t e = base + j;
      if (e < e1) {
#pragma unroll
        for (int k = 0; k < svp::kDog; ++k)
          if (segFirst[k] == e) ctl->segStart[k] = at + 1u;
      }
      if (keep[j]) {
        if (at < cap) {
          const uint32_t i = e / maxO;
          ssrlcv_sskeypoint kp = src[i];
          kp.theta = thetas[(size_t)i * svp::kMaxOrient + (e - i * maxO)];
          dst[at] = kp;
        }
        ++at;
      }
    }
    if (tile == tiles - 1 && threadIdx.x == 0) ctl->groupSum[g] = prefix[0] + total[0];
  }
  // bookkeeping: after EVERY block has left its loop -- a block that starts late must still read the old n
  __shared__ bool s_last;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    s_last = atomicAdd(&ctl->done[g], 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (s_last && threadIdx.x == 0) {
    __threadfence();
    const uint32_t total = before + __hip_atomic_load(&ctl->groupSum[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ctl->totals[g + 1] = total;  // where the next piece continues (book_orient's running sum)
    if (g + 1 == pieces && st->hasExtrema) {
      for (int b = 0; b < svp::kDog; ++b) {
        const uint32_t enc = __hip_atomic_load(&ctl->segStart[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t start = enc ? enc - 1u : total;  // a segment nobody met lies behind the last element
        st->idx[b] = (int)(start < cap ? start : cap);
      }
      uint32_t kept = total;
      if (kept > cap) { st->overflow = 1; kept = cap; }  // truncated: what lies past cap was not written
      st->n = (int)kept;
      if (kept == 0) st->hasExtrema = 0;
    }
  }
}

// the plan's 4 KB group block: four orientation work lists, four descriptor work lists, one GroupCtl per octave
struct GroupBlock {
  RangeTable thetaTab[svp::kSampleGroups];
  RangeTable descTab[svp::kSampleGroups];
  GroupCtl ctl[svp::kOctaves];
};
static_assert(sizeof(RangeTable) <= 256 && 8 * 256 + svp::kOctaves * sizeof(GroupCtl) <= 4096, "fits the plan's group block");
inline RangeTable* theta_tab(const ssrlcv_sift_plan* plan, char* ws, int g) { return (RangeTable*)(ws + plan->off_groups + 256 * g); }
inline RangeTable* desc_tab(const ssrlcv_sift_plan* plan, char* ws, int g) { return (RangeTable*)(ws + plan->off_groups + 256 * (svp::kSampleGroups + g)); }
inline GroupCtl* group_ctl(const ssrlcv_sift_plan* plan, char* ws, int o) { return (GroupCtl*)(ws + plan->off_groups + 2048) + o; }

// lanes per key point of the orientation kernel, as a shift (see k_thetas).  Developer build: SSRLCV_THETAS_LANES=1|2|4.
#ifndef SSRLCV_THETAS_LANES_SHIFT
#define SSRLCV_THETAS_LANES_SHIFT 2
#endif
inline int thetas_lanes_shift() {
  static const int shift = [] {
    const char* e = svdev::env("SSRLCV_THETAS_LANES");
    if (!e) return SSRLCV_THETAS_LANES_SHIFT;
    const int lanes = atoi(e);
    return lanes >= 4 ? 2 : (lanes >= 2 ? 1 : 0);
  }();
  return shift;
}
// upper bound of the orientation kernel's grid: one block per 64 >> shift key points of a range
inline unsigned thetas_grid(const ssrlcv_sift_plan* plan, int oFirst, int oLast, int shift) {
  unsigned blocks = 20;
  const uint32_t per = 64u >> shift;
  for (int o = oFirst; o <= oLast; ++o) blocks += (plan->oct[o].cap + per - 1) / per;
  return blocks;
}
template <int LPKS, typename... Args>
void launch_thetas_l(uint32_t maxO, unsigned blocks, hipStream_t st, Args... args) {
  switch (maxO) {
    case 1: hipLaunchKernelGGL((k_thetas<1, LPKS>), dim3(blocks), dim3(64), 0, st, args...); break;
    case 2: hipLaunchKernelGGL((k_thetas<2, LPKS>), dim3(blocks), dim3(64), 0, st, args...); break;
    case 3: hipLaunchKernelGGL((k_thetas<3, LPKS>), dim3(blocks), dim3(64), 0, st, args...); break;
    default: hipLaunchKernelGGL((k_thetas<4, LPKS>), dim3(blocks), dim3(64), 0, st, args...); break;
  }
}
template <typename... Args>
void launch_thetas(uint32_t maxO, int shift, unsigned blocks, hipStream_t st, Args... args) {
  switch (shift) {
    case 0: launch_thetas_l<0>(maxO, blocks, st, args...); break;
    case 1: launch_thetas_l<1>(maxO, blocks, st, args...); break;
    default: launch_thetas_l<2>(maxO, blocks, st, args...); break;
  }
}

// piece g of `pieces` of octave o's expansion (thrust::remove of the -FLT_MAX / -1 slots + expandKeyPoints, :594-611) on
// stream es; the tile descriptors of piece g live in its own slice of the octave's partition scratch
// (scratchZeroed: expand_prepare has cleared the piece's tile descriptors on some stream `es` is ordered behind)
int expand_prepare(const ssrlcv_sift_plan* plan, char* ws, int o, int g, hipStream_t st) {
  const svp::OctavePlan& oc = plan->oct[o];
  const uint32_t capTiles = (oc.cap * plan->params.maxOrientations + svs::kThreads * kExpandItems - 1) / (svs::kThreads * kExpandItems);
  const size_t scanBytes = svs::workspace_bytes<1>(capTiles);
  SSRLCV_HIP_TRY(hipMemsetAsync(ws + oc.off_part + (size_t)g * scanBytes, 0, scanBytes, st));
  return SSRLCV_OK;
}
int launch_expand(const ssrlcv_sift_plan* plan, char* ws, int o, int g, int pieces, hipStream_t es, bool scratchZeroed = false) {
  const svp::OctavePlan& oc = plan->oct[o];
  const uint32_t maxO = plan->params.maxOrientations, cap = oc.cap;
  OctaveState* states = (OctaveState*)(ws + plan->off_state);
  const ssrlcv_sskeypoint* src = (const ssrlcv_sskeypoint*)(ws + (plan->listInB[o] ? oc.off_kpB : oc.off_kpA));
  ssrlcv_sskeypoint* dst = (ssrlcv_sskeypoint*)(ws + (plan->listInB[o] ? oc.off_kpA : oc.off_kpB));
  const uint32_t capTiles = (cap * maxO + svs::kThreads * kExpandItems - 1) / (svs::kThreads * kExpandItems);
  const size_t scanBytes = svs::workspace_bytes<1>(capTiles);
  char* part = ws + oc.off_part + (size_t)g * scanBytes;
  if (!scratchZeroed) SSRLCV_HIP_TRY(hipMemsetAsync(part, 0, scanBytes, es));
  // a grid for the usual list lengths (a tenth of the capacity), persistent over the tiles
  unsigned blocks = capTiles / 8 + 1;
  blocks = blocks > 1024u ? 1024u : blocks;
  hipLaunchKernelGGL(k_expand_orient, dim3(blocks), dim3(svs::kThreads), 0, es, states + o, src, dst, (const float*)(ws + oc.off_theta),
                     (const uint32_t*)(ws + oc.off_thetaCnt), maxO, cap, svs::make_tile_scan<1>(part, capTiles), group_ctl(plan, ws, o), g, pieces);
  return SSRLCV_OK;
}
// the partition scratch of an octave holds `pieces` tile-descriptor slices
bool expand_scratch_fits(const ssrlcv_sift_plan* plan, int o, int pieces) {
  const svp::OctavePlan& oc = plan->oct[o];
  const uint32_t capTiles = (oc.cap * plan->params.maxOrientations + svs::kThreads * kExpandItems - 1) / (svs::kThreads * kExpandItems);
  const size_t P = (size_t)oc.w * oc.h;
  const size_t words = (size_t)3 * 4 * ((P + 8191) / 8192) + 16 + (size_t)svp::kDog * 4 * (((size_t)oc.cap * svp::kMaxOrient + 2047) / 2048) + 16;
  return (size_t)pieces * svs::workspace_bytes<1>(capTiles) <= words * 4;
}

// computeKeyPointOrientations (src/FeatureFactory.cu:540-632) for all octaves: gradient tables (unless the caller built
// them on a side stream already), one orientation launch, the expansion of every key point into its orientations
// Round 6 experiment (SSRLCV_THETAS_SPLIT=1): the orientation kernel's run time is largely that of its longest windows (a
// blur-3 key point alone takes 0.2 ms, whatever the octave), and octave 0's tables are three quarters of k_polar.  So
// describe_impl builds the small octaves' tables first and queues their orientations and expansions on a side stream `st`
// (behind their list chains and tables), beside octave 0's tables; octave 0's orientation launch follows its tables on the
// caller's stream (stage_orientations, restInFlight).  The caller has zeroed the octaves' GroupCtl.  Exact, and no faster:
// what runs beside the table kernel slows it by what it takes.
int queue_rest_orientations(const ssrlcv_sift_plan* plan, char* ws, hipStream_t st, svp::PlanAsync* as) {
  OctaveState* states = (OctaveState*)(ws + plan->off_state);
  const uint32_t maxO = plan->params.maxOrientations;
  OctaveSet set = make_set(plan, ws);
  uint32_t thetaSel = 0xFFFFFu;
#ifdef SSRLCV_INSTRUMENTED_BUILD
  if (const char* e = svdev::env("SSRLCV_TIMING_THETAS_SEL")) thetaSel = (uint32_t)strtoul(e, nullptr, 16);
#endif
  const int lanesShift = thetas_lanes_shift();
  RangeTable* restRanges = theta_tab(plan, ws, 1);
  hipLaunchKernelGGL(k_build_ranges, dim3(1), dim3(1), 0, st, states, restRanges, 6 - lanesShift, 0xFFFFFu & ~0x1Fu & thetaSel);
  launch_thetas(maxO, lanesShift, thetas_grid(plan, 1, svp::kOctaves - 1, lanesShift), st, (const OctaveState*)states, (const RangeTable*)restRanges,
                set, plan->params.orientationContribWidth, plan->params.orientationThreshold);
  for (int o = 1; o < svp::kOctaves; ++o) {
    int rc = launch_expand(plan, ws, o, 0, 1, st);
    if (rc) return rc;
    plan->listInB[o] ^= 1;
  }
  SSRLCV_HIP_TRY(hipEventRecord(as->expandJoin[1], st));
  return SSRLCV_OK;
}

// the stream octave o's expansion runs on
inline hipStream_t expand_stream(svp::PlanAsync* as, hipStream_t caller, int o) {
  return !as || o == 0 ? caller : (o == 1 ? as->chain : (o == 2 ? as->table : as->chain2));
}
// What the orientation stage can do before the gradient tables are complete (describe_impl calls it behind the list chains,
// in front of its wait for the table stream; round 6): the control words, the orientation work list, and -- on the streams
// the expansions will run on -- the clearing of their tile descriptors.
int orientations_prologue(const ssrlcv_sift_plan* plan, char* ws, hipStream_t caller, svp::PlanAsync* as) {
  OctaveState* states = (OctaveState*)(ws + plan->off_state);
  uint32_t thetaSel = 0xFFFFFu;
#ifdef SSRLCV_INSTRUMENTED_BUILD
  if (const char* e = svdev::env("SSRLCV_TIMING_THETAS_SEL")) thetaSel = (uint32_t)strtoul(e, nullptr, 16);
#endif
  SSRLCV_HIP_TRY(hipMemsetAsync(group_ctl(plan, ws, 0), 0, sizeof(GroupCtl) * svp::kOctaves, caller));
  hipLaunchKernelGGL(k_build_ranges, dim3(1), dim3(1), 0, caller, states, theta_tab(plan, ws, 0), 6 - thetas_lanes_shift(), thetaSel);
  if (as) {
    SSRLCV_HIP_TRY(hipEventRecord(as->expandFork, caller));
    for (int o = 1; o < svp::kOctaves; ++o) SSRLCV_HIP_TRY(hipStreamWaitEvent(expand_stream(as, caller, o), as->expandFork, 0));
  }
  for (int o = 0; o < svp::kOctaves; ++o) {
    const int rc = expand_prepare(plan, ws, o, 0, expand_stream(as, caller, o));
    if (rc) return rc;
  }
  return SSRLCV_OK;
}

int stage_orientations(const ssrlcv_sift_plan* plan, char* ws, hipStream_t caller, svp::PlanAsync* as, bool polarDone, bool restInFlight = false,
                       bool prologueDone = false) {
  OctaveState* states = (OctaveState*)(ws + plan->off_state);
  const uint32_t maxO = plan->params.maxOrientations;
  OctaveSet set = make_set(plan, ws);
  RangeTable* thetaRanges = theta_tab(plan, ws, 0);
  uint32_t thetaSel = 0xFFFFFu;
#ifdef SSRLCV_INSTRUMENTED_BUILD
  // timing only (results not valid): orientations of the selected (octave, blur segment) ranges alone -- bit o * 5 + seg
  if (const char* e = svdev::env("SSRLCV_TIMING_THETAS_SEL")) thetaSel = (uint32_t)strtoul(e, nullptr, 16);
#endif
  const int lanesShift = thetas_lanes_shift();
  const float lambdaO = plan->params.orientationContribWidth, othr = plan->params.orientationThreshold;
  // (Round 3 built this kernel with two and with four lanes per key point -- the 36 bins split between the lanes of a
  // group, each bin keeping its sequential chain, the samples' weights shared by DPP: bit-identical, and the same 0.58 ms
  // per 4096^2 image with 1, 2 or 4 lanes.  Round 6's form -- one histogram, the lanes taking turns -- is k_thetas' LPKS.)
  if (restInFlight) {
    // describe_impl has queued the small octaves' orientations and expansions on a side stream (queue_rest_orientations):
    // octave 0's follow its tables here
    hipLaunchKernelGGL(k_build_ranges, dim3(1), dim3(1), 0, caller, states, thetaRanges, 6 - lanesShift, 0x1Fu & thetaSel);
    launch_thetas(maxO, lanesShift, thetas_grid(plan, 0, 0, lanesShift), caller, (const OctaveState*)states, (const RangeTable*)thetaRanges, set,
                  lambdaO, othr);
    int rc = launch_expand(plan, ws, 0, 0, 1, caller);
    if (rc) return rc;
    plan->listInB[0] ^= 1;
    SSRLCV_HIP_TRY(hipStreamWaitEvent(caller, as->expandJoin[1], 0));
    return SSRLCV_OK;
  }
  if (!prologueDone) orientations_prologue(plan, ws, caller, as);
  if (!polarDone) launch_polar(plan, ws, caller);
  launch_thetas(maxO, lanesShift, thetas_grid(plan, 0, svp::kOctaves - 1, lanesShift), caller, (const OctaveState*)states, (const RangeTable*)thetaRanges, set, lambdaO, othr);
  // The four expansions are independent chains of ~30 us each (launch-bound on the short lists): those of octaves 1-3 run on
  // the three side streams beside octave 0's on the caller's stream (their tile descriptors were cleared by the prologue).
  if (as) {
    SSRLCV_HIP_TRY(hipEventRecord(as->groupFork, caller));
    for (int o = 1; o < svp::kOctaves; ++o) SSRLCV_HIP_TRY(hipStreamWaitEvent(expand_stream(as, caller, o), as->groupFork, 0));
  }
  for (int o = 0; o < svp::kOctaves; ++o) {
    int rc = launch_expand(plan, ws, o, 0, 1, expand_stream(as, caller, o), true);
    if (rc) return rc;
    plan->listInB[o] ^= 1;
  }
  if (as) {
    for (int o = 1; o < svp::kOctaves; ++o) {
      SSRLCV_HIP_TRY(hipEventRecord(as->groupExpanded[o - 1], expand_stream(as, caller, o)));
      SSRLCV_HIP_TRY(hipStreamWaitEvent(caller, as->groupExpanded[o - 1], 0));
    }
  }
  return SSRLCV_OK;
}

// withDescRanges: also the descriptor work list of all octaves (what stage_descriptors would launch k_build_ranges for)
void book_features(const ssrlcv_sift_plan* plan, char* ws, uint32_t* numFeatures, hipStream_t caller, bool withDescRanges = false) {
  if (withDescRanges)
    hipLaunchKernelGGL(k_book_and_desc_ranges, dim3(1), dim3(1), 0, caller, (OctaveState*)(ws + plan->off_state),
                       (uint32_t*)(ws + plan->oct[0].off_featBase), numFeatures, plan->maxFeatures, desc_tab(plan, ws, 0));
  else
    hipLaunchKernelGGL(k_book_featbase, dim3(1), dim3(1), 0, caller, (OctaveState*)(ws + plan->off_state),
                       (uint32_t*)(ws + plan->oct[0].off_featBase), numFeatures, plan->maxFeatures);
}

unsigned desc_blocks(const ssrlcv_sift_plan* plan, int oFirst, int oLast, unsigned* maxBlocks) {
  unsigned blocks = 0, mx = 1;
  for (int o = oFirst; o <= oLast; ++o) {
    blocks += list_blocks(plan->oct[o].cap);
    mx = list_blocks(plan->oct[o].cap) > mx ? list_blocks(plan->oct[o].cap) : mx;
  }
  if (maxBlocks) *maxBlocks = mx;
  return blocks;
}

// fillDescriptors (src/SIFT_FeatureFactory.cu:131-166,475-549), all octaves in one launch (book_features first)
int stage_descriptors(const ssrlcv_sift_plan* plan, char* ws, ssrlcv_sift_feature* features, hipStream_t caller, bool rangesBuilt = false) {
  OctaveState* states = (OctaveState*)(ws + plan->off_state);
  OctaveSet set = make_set(plan, ws);
  RangeTable* descRanges = desc_tab(plan, ws, 0);
  uint32_t* featBase = (uint32_t*)(ws + plan->oct[0].off_featBase);
  unsigned maxBlocks = 1;
  const unsigned descBlocks = desc_blocks(plan, 0, svp::kOctaves - 1, &maxBlocks);
  // one launch over every octave's key points, largest windows first (see RangeTable)
  if (!rangesBuilt) hipLaunchKernelGGL(k_build_ranges, dim3(1), dim3(1), 0, caller, states, descRanges, 0, 0xFFFFFu);
  hipLaunchKernelGGL(k_desc_consts, dim3(maxBlocks * svp::kOctaves), dim3(256), 0, caller, (const RangeTable*)descRanges, set,
                     plan->params.descriptorContribWidth, (const OctaveState*)states, (const GroupCtl*)nullptr);
  hipLaunchKernelGGL(k_descriptors, dim3(descBlocks * kWaveKernelOversubscription), dim3(256), 0, caller, (const RangeTable*)descRanges, set,
                     (const uint32_t*)featBase, features, plan->maxFeatures);
  return SSRLCV_OK;
}

// computeKeyPointOrientations + fillDescriptors pipelined over the sampling groups (see k_build_group_ranges): the
// orientation launches of all groups start together (group 0 on the caller's stream, the others on the side streams); a
// group's expansion, constants and work list follow on its stream, and the caller's stream runs the four descriptor
// launches in output order, each behind its group's `ready` event.  Results are those of stage_orientations +
// stage_descriptors: the same kernels over the same key points, the expansion in pieces that continue each other.
int stage_sampling_pipelined(const ssrlcv_sift_plan* plan, char* ws, ssrlcv_sift_feature* features, uint32_t* numFeatures,
                             hipStream_t caller, svp::PlanAsync* as) {
  OctaveState* states = (OctaveState*)(ws + plan->off_state);
  const uint32_t maxO = plan->params.maxOrientations;
  const float lambdaO = plan->params.orientationContribWidth, othr = plan->params.orientationThreshold;
  const float lambdaD = plan->params.descriptorContribWidth;
  uint32_t* featBase = (uint32_t*)(ws + plan->oct[0].off_featBase);
  OctaveSet setT = make_set(plan, ws);  // the lists as the chains left them: what k_thetas reads
  for (int o = 0; o < svp::kOctaves; ++o) plan->listInB[o] ^= 1;
  OctaveSet setD = make_set(plan, ws);  // the expanded lists
  for (int o = 0; o < svp::kOctaves; ++o) plan->listInB[o] ^= 1;  // (launch_expand reads the un-flipped state)
  const hipStream_t gs[svp::kSampleGroups] = {caller, as->chain, as->chain2, as->table};
  const int lanesShift = thetas_lanes_shift();
  unsigned thetaBlocks[svp::kSampleGroups], descBlocks[svp::kSampleGroups], constBlocks[svp::kSampleGroups];
  for (int g = 0; g < svp::kSampleGroups; ++g) {
    const int oFirst = g < kOct0Groups ? 0 : 1, oLast = g < kOct0Groups ? 0 : svp::kOctaves - 1;
    thetaBlocks[g] = thetas_grid(plan, oFirst, oLast, lanesShift);
    unsigned mx = 1;
    descBlocks[g] = desc_blocks(plan, oFirst, oLast, &mx) * kWaveKernelOversubscription;
    constBlocks[g] = mx * (unsigned)(oLast - oFirst + 1);
  }
  SSRLCV_HIP_TRY(hipMemsetAsync(group_ctl(plan, ws, 0), 0, sizeof(GroupCtl) * svp::kOctaves, caller));
  SSRLCV_HIP_TRY(hipMemsetAsync(featBase, 0, 4 * svp::kOctaves, caller));  // octave 0's features start at 0, known now
  // (developer build: SSRLCV_SAMPLING_IRREGULAR=1 takes the fallback for blur indices that are not an ordered partition)
  static const bool forceIrregular = svdev::env("SSRLCV_SAMPLING_IRREGULAR") != nullptr;
  hipLaunchKernelGGL(k_build_group_ranges, dim3(1), dim3(1), 0, caller, (const OctaveState*)states, theta_tab(plan, ws, 0), group_ctl(plan, ws, 0), 6 - lanesShift,
                     forceIrregular);
  SSRLCV_HIP_TRY(hipEventRecord(as->groupFork, caller));
  for (int g = 1; g < svp::kSampleGroups; ++g) SSRLCV_HIP_TRY(hipStreamWaitEvent(gs[g], as->groupFork, 0));
  for (int g = 0; g < svp::kSampleGroups; ++g)
    launch_thetas(maxO, lanesShift, thetaBlocks[g], gs[g], (const OctaveState*)states, (const RangeTable*)theta_tab(plan, ws, g), setT, lambdaO, othr);
  // octave 0: three pieces of one expansion, piece g behind piece g - 1
  for (int g = 0; g < kOct0Groups; ++g) {
    if (g > 0) SSRLCV_HIP_TRY(hipStreamWaitEvent(gs[g], as->groupExpanded[g - 1], 0));
    int rc = launch_expand(plan, ws, 0, g, kOct0Groups, gs[g]);
    if (rc) return rc;
    SSRLCV_HIP_TRY(hipEventRecord(as->groupExpanded[g], gs[g]));
    hipLaunchKernelGGL(k_build_desc_ranges_oct0, dim3(1), dim3(1), 0, gs[g], (const OctaveState*)states, desc_tab(plan, ws, g),
                       (const GroupCtl*)group_ctl(plan, ws, 0), g, plan->oct[0].cap);
    hipLaunchKernelGGL(k_desc_consts, dim3(constBlocks[g]), dim3(256), 0, gs[g], (const RangeTable*)desc_tab(plan, ws, g), setD, lambdaD,
                       (const OctaveState*)states, (const GroupCtl*)group_ctl(plan, ws, 0));
    if (g > 0) SSRLCV_HIP_TRY(hipEventRecord(as->groupReady[g], gs[g]));
  }
  // octaves 1-3: whole-octave expansions, then the feature offsets (octave 0's final count: piece 2), constants, work list
  {
    const int g = svp::kSampleGroups - 1;
    for (int o = 1; o < svp::kOctaves; ++o) {
      int rc = launch_expand(plan, ws, o, 0, 1, gs[g]);
      if (rc) return rc;
    }
    SSRLCV_HIP_TRY(hipStreamWaitEvent(gs[g], as->groupExpanded[kOct0Groups - 1], 0));
    book_features(plan, ws, numFeatures, gs[g]);
    hipLaunchKernelGGL(k_build_ranges, dim3(1), dim3(1), 0, gs[g], (const OctaveState*)states, desc_tab(plan, ws, g), 0, 0xFFFFFu & ~0x1Fu);
    hipLaunchKernelGGL(k_desc_consts, dim3(constBlocks[g]), dim3(256), 0, gs[g], (const RangeTable*)desc_tab(plan, ws, g), setD, lambdaD,
                       (const OctaveState*)states, (const GroupCtl*)nullptr);
    SSRLCV_HIP_TRY(hipEventRecord(as->groupReady[g], gs[g]));
  }
  for (int o = 0; o < svp::kOctaves; ++o) plan->listInB[o] ^= 1;
  for (int g = 0; g < svp::kSampleGroups; ++g) {
    if (g > 0) SSRLCV_HIP_TRY(hipStreamWaitEvent(caller, as->groupReady[g], 0));
    hipLaunchKernelGGL(k_descriptors, dim3(descBlocks[g]), dim3(256), 0, caller, (const RangeTable*)desc_tab(plan, ws, g), setD,
                       (const uint32_t*)featBase, features, plan->maxFeatures);
  }
  return SSRLCV_OK;
}

// the list chain of one octave (S8 tail - S12) up to the plan's stop stage, on `s`
int run_list_chain(const ssrlcv_sift_plan* plan, char* ws, int o, hipStream_t s) {
  const int stop = plan->stopStage;
  const ListCtx c = make_ctx(plan, ws, o, s);
  int rc = stage_extrema(c, stop >= 1);
  if (!rc && stop >= 2) rc = stage_refine(c);
  // stages 3-5 (noise, edges, window check) collapse into one discard when the run goes past them: the three tests are
  // independent per key point and the compaction is stable
  if (!rc && stop >= 5) rc = stage_noise_edges_window(c);
  else {
    if (!rc && stop >= 3) rc = stage_noise(c, svp::kNoiseThreshold);
    if (!rc && stop >= 4) rc = stage_edges(c);
  }
  return rc;
}

}  // namespace
namespace svp {
int launch_chain_octave(const ssrlcv_sift_plan* plan, char* ws, int octave, hipStream_t st) { return run_list_chain(plan, ws, octave, st); }
}  // namespace svp
extern "C" {

// A stage call that fails half-way may have left work queued on the plan's side streams (octave 0's list chain on `chain2`,
// the gradient tables on `polar` / `table`, queued by build_dog of a fused extract, or the chains this call forked): the
// caller's stream is ordered behind all of it and the in-flight marks are cleared, so that whatever the caller does next
// with the workspace (another extract, a stand-alone describe, freeing it behind a stream synchronisation) is ordered
// behind those writes.  Errors of the drain itself are ignored: the call already has one to report.
static void drain_side_streams(const ssrlcv_sift_plan* plan, hipStream_t st) {
  svp::PlanAsync* as = plan->asyncState == 1 ? plan->async : nullptr;
  if (as) {
    for (hipStream_t side : {as->chain, as->table, as->chain2, as->polar}) {
      if (hipEventRecord(as->fork, side) == hipSuccess) (void)hipStreamWaitEvent(st, as->fork, 0);
    }
    (void)hipGetLastError();
  }
  plan->chain0InFlight = 0;
  plan->polarInFlight = 0;
}
static int describe_impl(const ssrlcv_sift_plan* plan, void* workspace, ssrlcv_sift_feature* features, uint32_t* numFeatures, ssrlcv_stream_t stream);
int ssrlcv_hip_sift_describe(const ssrlcv_sift_plan* plan, void* workspace, ssrlcv_sift_feature* features, uint32_t* numFeatures,
                             ssrlcv_stream_t stream) {
  const int rc = describe_impl(plan, workspace, features, numFeatures, stream);
  if (rc && plan) drain_side_streams(plan, (hipStream_t)stream);
  return rc;
}
static int describe_impl(const ssrlcv_sift_plan* plan, void* workspace, ssrlcv_sift_feature* features,
                             uint32_t* numFeatures, ssrlcv_stream_t stream) {
  if (!plan || !workspace || !numFeatures) return SSRLCV_ERR_INVALID_ARG;
  char* ws = (char*)workspace;
  const int stop = plan->stopStage;
  if (stop >= 7 && !features) return SSRLCV_ERR_INVALID_ARG;
  // The four octaves' chains are independent until the feature offsets are summed, and each is a long run of small
  // launches (bookkeeping kernels, compactions of short lists): octave 0's runs on the caller's stream, octave 1's on
  // `chain`, those of octaves 2-3 on `chain2`, the polar tables on `table`, all forked from and joined back into the
  // caller's stream.
  svp::PlanAsync* as = svp::plan_async(plan);
  const hipStream_t caller = (hipStream_t)stream;
  // Developer build, SSRLCV_THETAS_SPLIT=1 (round 6; exact, measured, NOT the default -- describe 3.09-3.11 ms per image either
  // way, profiles/r06_kernel_ab.txt): the small octaves' tables first and their orientations beside octave 0's tables.
  static const bool splitWanted = svdev::env("SSRLCV_THETAS_SPLIT") != nullptr;
  static const bool pipelinedSampling = svdev::env("SSRLCV_SAMPLING_PIPELINED") != nullptr;
  const bool splitRest = as && stop >= 6 && !plan->polarInFlight && splitWanted && !pipelinedSampling;
  if (splitRest) SSRLCV_HIP_TRY(hipMemsetAsync(group_ctl(plan, ws, 0), 0, sizeof(GroupCtl) * svp::kOctaves, caller));
  if (as) {
    SSRLCV_HIP_TRY(hipEventRecord(as->fork, caller));
    SSRLCV_HIP_TRY(hipStreamWaitEvent(as->chain, as->fork, 0));
    SSRLCV_HIP_TRY(hipStreamWaitEvent(as->chain2, as->fork, 0));
    SSRLCV_HIP_TRY(hipStreamWaitEvent(as->table, as->fork, 0));
    if ((svp::stream_priority_mode() & 1)) SSRLCV_HIP_TRY(hipStreamWaitEvent(as->polar, as->fork, 0));
    if (stop >= 6 && !plan->polarInFlight) {
      const hipStream_t ps = (svp::stream_priority_mode() & 1) ? as->polar : as->table;
      if (splitRest) {  // the small octaves' tables first: their orientations start behind them (below)
        launch_polar(plan, ws, ps, 1, svp::kOctaves - 1);
        SSRLCV_HIP_TRY(hipEventRecord(as->polarDone[1], ps));
        launch_polar(plan, ws, ps, 0, 0);
      } else {
        launch_polar(plan, ws, ps);
      }
    }
  }
  // (Round 5 built the chains of all four octaves as five launches on one stream -- every partition one decoupled look-back
  // pass, its class totals counted by the launch before it, the bookkeeping in a thread of the launch that produced its inputs;
  // exact, and slower: 0.58-0.73 ms alone on a 4096^2 image against 0.43 for the staged chains below, whose wave-granular
  // count -> scan -> scatter kernels keep a key point per thread in flight where a look-back tile serialises several per
  // thread behind block-wide scans.  Removed; it is commit "Experiment: the list chains of all octaves ..." in the history,
  // numbers in profiles/r05_schedule_ab.txt.)
  for (int o = 0; o < svp::kOctaves; ++o) {
    // octave 1's chain on one side stream, those of octaves 2 and 3 one after the other on a second (round 3: the three
    // short chains in a row on one stream, ~25 launch-bound kernels each, ended after the polar tables)
    const hipStream_t cs = !as || o == 0 ? caller : (o == 1 ? as->chain : as->chain2);
    // (bit o of chain0InFlight: the octave's chain is already in flight on `chain2`, queued by build_dog of the same fused extract
    // behind that octave's DoG pass: the in-order stream runs it in front of the chains of octaves 2-3, whose join event covers it)
    if (!((plan->chain0InFlight >> o) & 1)) {
      const int rc = run_list_chain(plan, ws, o, cs);
      if (rc) return rc;
    }
    if (as && (o == 1 || o == svp::kOctaves - 1)) {  // the side streams are in order: their last events join them
      SSRLCV_HIP_TRY(hipEventRecord(as->join[o], cs));
      SSRLCV_HIP_TRY(hipStreamWaitEvent(caller, as->join[o], 0));
    }
  }
  plan->chain0InFlight = 0;
  if (splitRest) {  // behind the chains of octaves 2-3 on their stream: octave 1's chain, the small octaves' tables, then their orientations
    SSRLCV_HIP_TRY(hipStreamWaitEvent(as->chain2, as->join[1], 0));
    SSRLCV_HIP_TRY(hipStreamWaitEvent(as->chain2, as->polarDone[1], 0));
    const int rc = queue_rest_orientations(plan, ws, as->chain2, as);
    if (rc) return rc;
  }
  const bool earlyPrologue = as && stop >= 6 && !splitRest && !(stop >= 7 && pipelinedSampling && expand_scratch_fits(plan, 0, kOct0Groups));
  if (earlyPrologue) {
    const int rc = orientations_prologue(plan, ws, caller, as);
    if (rc) return rc;
  }
  if (as) {  // the polar stream joins too (its tables are read by the sampling kernels below)
    SSRLCV_HIP_TRY(hipEventRecord(as->join[svp::kOctaves], (svp::stream_priority_mode() & 1) ? as->polar : as->table));
    SSRLCV_HIP_TRY(hipStreamWaitEvent(caller, as->join[svp::kOctaves], 0));
    if (plan->polarInFlight) {  // queued by build_dog of the same fused extract (one launch per octave on `polar`)
      for (int o = 0; o < svp::kOctaves; ++o) SSRLCV_HIP_TRY(hipStreamWaitEvent(caller, as->polarDone[o], 0));
      plan->polarInFlight = 0;
    }
  }
  // Developer build, SSRLCV_SAMPLING_PIPELINED=1: the pipelined form (orientation launches of the later sampling groups
  // beside the descriptor launches of the earlier ones).  Exact (tests/test_gpu_sift.py), measured in round 5 and NOT the
  // default: the step took 10.29 ms with it against 10.19 without (profiles/r05_schedule_ab.txt) -- the resident one-wave
  // orientation blocks hold 9 KB of LDS each, 13 per CU, and the descriptor blocks beside them run at a quarter of their
  // occupancy; both kernels also draw on the same gather bandwidth of the polar tables.
  if (stop >= 7 && as && pipelinedSampling && expand_scratch_fits(plan, 0, kOct0Groups)) {
    int rc = stage_sampling_pipelined(plan, ws, features, numFeatures, caller, as);
    if (rc) return rc;
    SSRLCV_LAUNCH_CHECK();
    return SSRLCV_OK;
  }
  if (stop >= 6) {
    int rc = stage_orientations(plan, ws, caller, as, as != nullptr, splitRest, earlyPrologue);
    if (rc) return rc;
  }
  book_features(plan, ws, numFeatures, caller, stop >= 7);
  if (stop >= 7) {
    int rc = stage_descriptors(plan, ws, features, caller, true);
    if (rc) return rc;
  }
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

// One reference launch site at a time (INTEGRATION.md option B) on the state the previous stages left in `workspace`:
//   0 searchForExtrema  1 removeNoise(0.8 x 0.01)  2 refineExtremaLocation (+ sort, re-scan)  3 removeNoise(0.01)
//   4 removeEdges  5 checkKeyPoints  6 computeKeyPointOrientations  7 fillDescriptors
int ssrlcv_hip_sift_stage(const ssrlcv_sift_plan* plan, void* workspace, int stage, ssrlcv_sift_feature* features,
                          uint32_t* numFeatures, ssrlcv_stream_t stream) {
  if (!plan || !workspace || !numFeatures || stage < 0 || stage > 7) return SSRLCV_ERR_INVALID_ARG;
  if (stage == 7 && !features) return SSRLCV_ERR_INVALID_ARG;
  char* ws = (char*)workspace;
  const hipStream_t caller = (hipStream_t)stream;
  int rc = SSRLCV_OK;
  if (stage <= 5) {
    for (int o = 0; o < svp::kOctaves && !rc; ++o) {
      const ListCtx c = make_ctx(plan, ws, o, caller);
      switch (stage) {
        case 0: rc = stage_extrema(c, false); break;
        case 1: rc = stage_noise(c, (float)(svp::kNoiseThreshold * 0.8)); break;
        case 2: rc = stage_refine(c); break;
        case 3: rc = stage_noise(c, svp::kNoiseThreshold); break;
        case 4: rc = stage_edges(c); break;
        default: rc = stage_window(c); break;
      }
    }
  } else if (stage == 6) {
    rc = stage_orientations(plan, ws, caller, nullptr, false);
  }
  if (rc) return rc;
  book_features(plan, ws, numFeatures, caller);
  if (stage == 7) rc = stage_descriptors(plan, ws, features, caller);
  if (rc) return rc;
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

// =====================================================================================================================
// Per-kernel entry points over the CALLER's buffers (round 4): one call per launch site of
// FeatureFactory::ScaleSpace / SIFT_FeatureFactory, for a maintainer who keeps upstream's ScaleSpace objects (DoG images in
// Octave::blurs[b]->pixels, Unity<SSKeyPoint> lists, extremaBlurIndices on the host) and swaps kernels one at a time.
// These are the plain forms -- they read materialised DoG images and whole gradient arrays like the kernels they replace,
// not the plan's fused workspace -- with the plan path's arithmetic: same device functions (refine_one,
// edge_response_above, sv_math.h), the descriptor sums in the order-independent form of DESIGN.md section 2.
}  // extern "C"
namespace {
// findExtrema (src/FeatureFactory.cu:847-882): non-strict 3x3x3 maximum / minimum; border pixels are not written
__global__ __launch_bounds__(256) void k_x_find_extrema(int W, int H, const float* __restrict__ up, const float* __restrict__ mid,
                                                        const float* __restrict__ low, int* __restrict__ extrema) {
  const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= (size_t)W * H) return;
  const int x = (int)(id % W), y = (int)(id / W);
  if (!(x > 0 && y > 0 && x < W - 1 && y < H - 1)) return;
  const float v = mid[id];
  float mx = -FLT_MAX, mn = FLT_MAX;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const size_t a = (size_t)(y + dy) * W + (x + dx);
      mx = fmaxf(mx, fmaxf(low[a], fmaxf(mid[a], up[a])));
      mn = fminf(mn, fminf(low[a], fminf(mid[a], up[a])));
    }
  extrema[id] = (mx == v || mn == v) ? (int)id : -1;
}
// fillExtrema (:883-890)
__global__ __launch_bounds__(256) void k_x_fill_extrema(int n, int W, int octave, int blur, float sigma, const int* __restrict__ addr,
                                                        const float* __restrict__ pixels, ssrlcv_sskeypoint* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int index = addr[i];
  ssrlcv_sskeypoint kp;
  kp.octave = octave;
  kp.blur = blur;
  kp.loc.x = (float)(index % W);
  kp.loc.y = (float)(index / W);
  kp.intensity = pixels[index];
  kp.sigma = sigma;
  kp.theta = -1.0f;
  kp.discard = 0;
  out[i] = kp;
}
__global__ __launch_bounds__(256) void k_x_flag_noise(uint32_t n, ssrlcv_sskeypoint* kps, float thr) {  // :968-973
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) kps[i].discard = (uint8_t)(fabsf(kps[i].intensity) < thr);
}
__global__ __launch_bounds__(256) void k_x_flag_edges(uint32_t n, uint32_t start, int W, ssrlcv_sskeypoint* kps, const float* pixels,
                                                      float thr) {  // :974-990
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) kps[start + i].discard = (uint8_t)edge_response_above(MatView{pixels}, kps[start + i], W, thr);
}
__global__ __launch_bounds__(256) void k_x_check(uint32_t n, uint32_t start, int W, int H, float pixelWidth, float lambda,
                                                 ssrlcv_sskeypoint* kps) {  // src/SIFT_FeatureFactory.cu:449-461: only ever SETS the flag
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const ssrlcv_sskeypoint kp = kps[start + i];
  const float ww = kp.sigma * lambda / pixelWidth;
  if ((kp.loc.x - ww) < 0.0f || (kp.loc.y - ww) < 0.0f || (kp.loc.x + ww) >= (unsigned)(W - 1) || (kp.loc.y + ww) >= (unsigned)(H - 1))
    kps[start + i].discard = 1;
}
__global__ __launch_bounds__(256) void k_x_refine(uint32_t n, int W, int H, float sigmaMin, float mult, int numBlurs, MatSource src,
                                                  ssrlcv_sskeypoint* kps) {  // :892-967
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  ssrlcv_sskeypoint kp = kps[i];
  refine_one(kp, W, H, numBlurs, sigmaMin, mult, src);
  kps[i] = kp;
}
// calculatePixelGradients (src/Image.cu:1583-1598): a border pixel takes the stencil of its inner neighbour
__device__ __forceinline__ float2 x_gradient(const float* __restrict__ px, int W, int H, int x, int y) {
  int xc0 = x + 1, xc1 = x - 1, yc0 = y + 1, yc1 = y - 1;
  if (xc1 == -1) { xc0 += 1; xc1 += 1; }
  else if (xc0 == W) { xc0 -= 1; xc1 -= 1; }
  if (yc1 == -1) { yc0 += 1; yc1 += 1; }
  else if (yc0 == H) { yc0 -= 1; yc1 -= 1; }
  float2 g;
  g.x = px[(size_t)y * W + xc0] - px[(size_t)y * W + xc1];
  g.y = px[(size_t)yc0 * W + x] - px[(size_t)yc1 * W + x];
  return g;
}
__global__ __launch_bounds__(256) void k_x_gradients(int W, int H, const float* __restrict__ px, float2* __restrict__ grad) {
  const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (id < (size_t)W * H) grad[id] = x_gradient(px, W, H, (int)(id % W), (int)(id / W));
}
// computeThetas (src/FeatureFactory.cu:1004-1112): one thread per key point, the reference's own shape (its sequential
// fmaf chain per bin is part of the result).  thetas / thetaNumbers: n x maxOrientations, -FLT_MAX / -1 = none.
__global__ __launch_bounds__(64) void k_x_thetas(uint32_t n, uint32_t start, int W, int H, float pixelWidth, float lambda,
                                                 const ssrlcv_sskeypoint* __restrict__ kps, const float2* __restrict__ grad,
                                                 int* __restrict__ thetaNumbers, uint32_t maxO, float orientationThreshold,
                                                 float* __restrict__ thetas) {
  const uint32_t gi = blockIdx.x * 64 + threadIdx.x;
  if (gi >= n) return;
  const float pi = 3.1415927f;
  const ssrlcv_sskeypoint kp = kps[start + gi];
  const int regNumOrient = (int)(maxO > 8u ? 8u : maxO);
  for (int i = 0; i < regNumOrient; ++i) { thetas[(size_t)gi * maxO + i] = -FLT_MAX; thetaNumbers[(size_t)gi * maxO + i] = -1; }
  const float kx = kp.loc.x, ky = kp.loc.y;
  const float windowWidth = ceilf(kp.sigma * 3.0f * lambda / pixelWidth);
  const float minx = kx - windowWidth, miny = ky - windowWidth, maxx = kx + windowWidth, maxy = ky + windowWidth;
  if (minx < 0.0f || miny < 0.0f || maxx >= (unsigned)(W - 1) || maxy >= (unsigned)(H - 1)) return;
  float hist[36];
  for (int i = 0; i < 36; ++i) hist[i] = 0.0f;
  const float weight = 2.0f * lambda * lambda * kp.sigma * kp.sigma;
  const float rad10 = pi / 18.0f;
  for (float y = miny; y <= maxy; y += 1.0f) {
    for (float x = minx; x <= maxx; x += 1.0f) {
      const float2 g = grad[(size_t)llroundf(y) * W + (size_t)llroundf(x)];
      const float tx = x - kx, ty = y - ky;
      const float angle = fmodf(sv_atan2f(g.y, g.x) + (2.0f * pi), 2.0f * pi);
      const int bin = (int)floorf(angle / rad10);
      const float mag = sqrtf((g.x * g.x) + (g.y * g.y));
      const float wgt = sv_expf(-((tx * tx) + (ty * ty)) / weight);
      // dynamic index into the per-thread histogram: a select chain keeps it in registers
#pragma unroll
      for (int b = 0; b < 36; ++b)
        if (b == bin) hist[b] = __builtin_fmaf(mag, wgt, hist[b]);
    }
  }
  float maxHist = 0.0f;
  for (int i = 0; i < 36; ++i)
    if (hist[i] > maxHist) maxHist = hist[i];
  maxHist *= orientationThreshold;
  float bx[8], by[8];
  for (int i = 0; i < 8; ++i) { bx[i] = 0.0f; by[i] = 0.0f; }
  for (int b = 0; b < 36; ++b) {
    const float hb = hist[b], hp = hist[b == 0 ? 35 : b - 1], hn = hist[b == 35 ? 0 : b + 1];
    if (hb < maxHist || hb < hp || hb < hn || hb < bx[regNumOrient - 1]) continue;
    float tx = hb;
    float ty = (hp - hn) / (hp - (2.0f * hb) + hn);
    ty *= (pi / 36.0f);
    ty += (b * rad10);
    ty = fmodf(ty + (2.0f * pi), 2.0f * pi);
    for (int i = 0; i < regNumOrient; ++i) {
      if (tx > bx[i]) {
        for (int ii = i; ii < regNumOrient; ++ii) {
          const float sx = bx[ii], sy = by[ii];
          bx[ii] = tx;
          by[ii] = ty;
          tx = sx;
          ty = sy;
        }
      }
    }
  }
  for (int i = 0; i < regNumOrient; ++i) {
    if (bx[i] != 0.0f) {
      thetas[(size_t)gi * maxO + i] = by[i];
      thetaNumbers[(size_t)gi * maxO + i] = (int)(start + gi);
    }
  }
}
__global__ __launch_bounds__(256) void k_x_expand(uint32_t n, const ssrlcv_sskeypoint* __restrict__ in, ssrlcv_sskeypoint* __restrict__ out,
                                                  const int* __restrict__ thetaAddresses, const float* __restrict__ thetas) {  // :1114-1122
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  ssrlcv_sskeypoint kp = in[thetaAddresses[i]];
  kp.theta = thetas[i];
  out[i] = kp;
}
// fillDescriptors (src/SIFT_FeatureFactory.cu:475-549): one wave per key point; the votes are the reference's expressions,
// the 128 bins are summed as integers (vote x 2^k rounded half up: order independent), the norms as
// balanced trees -- the definition k_descriptors and the oracle share.
__global__ __launch_bounds__(64) void k_x_descriptors(uint32_t n, uint32_t start, int W, int H, float pixelWidth, float lambda,
                                                      const ssrlcv_sskeypoint* __restrict__ kps, const float2* __restrict__ grad,
                                                      ssrlcv_sift_feature* __restrict__ features) {
  __shared__ unsigned s_bins[128];
  const uint32_t gi = blockIdx.x;
  if (gi >= n) return;
  const int lane = threadIdx.x;
  const float pi = 3.1415927f;
  const ssrlcv_sskeypoint kp = kps[start + gi];
  s_bins[lane] = 0u;
  s_bins[lane + 64] = 0u;
  __syncthreads();
  const float kx = kp.loc.x, ky = kp.loc.y, theta = kp.theta;
  const float windowWidth = ceilf(kp.sigma * lambda / pixelWidth);
  const float binWidth = windowWidth / 2.0f, rad45 = pi / 4.0f;
  const float c = sv_cosf(-theta), s = sv_sinf(-theta);
  int boundExp;
  (void)frexpf(1.4143f * ((windowWidth + 2.0f) * (windowWidth + 2.0f)), &boundExp);
  const float voteScale = ldexpf(1.0f, 31 - boundExp);
  const int iw = (int)windowWidth, side = 2 * iw + 1;
  for (int sIdx = lane; sIdx < side * side; sIdx += 64) {
    const float y = (float)(sIdx / side - iw), x = (float)(sIdx % side - iw);
    const float cx = (x * c) + (y * s), cy = (-x * s) + (y * c);
    if (fabsf(cx) > windowWidth || fabsf(cy) > windowWidth) continue;
    const long long flat = llroundf(cy + ky) * (long long)W + llroundf(cx + kx);
    float2 g;
    g.x = g.y = 0.0f;
    if (flat >= 0 && flat < (long long)W * H) g = grad[flat];
    const float mag = sqrtf((g.x * g.x) + (g.y * g.y)) * sv_expf(-((cx * cx) + (cy * cy)) / (2.0f * windowWidth * windowWidth));
    const float ang = fmodf(sv_atan2f(g.y, g.x) - theta + (2.0f * pi), 2.0f * pi);
    for (int nx = 0; nx < 4; ++nx) {
      for (int ny = 0; ny < 4; ++ny) {
        float hx = ((float)nx * 0.5f - 0.75f) * windowWidth, hy = ((float)ny * 0.5f - 0.75f) * windowWidth;
        const float rx = (hx * c) + (hy * s), ry = (-hx * s) + (hy * c);
        hx = fabsf(rx - cx);
        hy = fabsf(ry - cy);
        if (hx <= binWidth && hy <= binWidth) {
          hx = hx / binWidth;
          hy = hy / binWidth;
          for (int k = 0; k < 8; ++k) {
            float angle = fabsf(ang - ((float)k * rad45));
            if (angle < rad45) {
              angle /= rad45;
              const float temp = (1.0f - hx) * (1.0f - hy) * (1.0f - angle) * mag;
              const float q = temp * voteScale, f = floorf(q);
              atomicAdd(&s_bins[(nx * 4 + ny) * 8 + k], (unsigned)f + ((q - f) >= 0.5f ? 1u : 0u));
            }
          }
        }
      }
    }
  }
  __syncthreads();
  // bins in [nx][ny][k] order: lane l holds elements l and l + 64; both norms as balanced trees (pairs 64 apart first)
  float v0 = (float)s_bins[lane], v1 = (float)s_bins[lane + 64];
  float sq = sqrtf(sv::wave_sum((v0 * v0) + (v1 * v1)));
  v0 /= sq;
  v1 /= sq;
  v0 = v0 > 0.2f ? 0.2f : v0;
  v1 = v1 > 0.2f ? 0.2f : v1;
  sq = sqrtf(sv::wave_sum((v0 * v0) + (v1 * v1)));
  ssrlcv_sift_feature* ft = features + gi;
  // values[(ny * 4 + nx) * 8 + k] = bin[nx][ny][k]
  {
    const int e0 = lane, e1 = lane + 64;
    const int nx0 = e0 >> 5, ny0 = (e0 >> 3) & 3, k0 = e0 & 7, nx1 = e1 >> 5, ny1 = (e1 >> 3) & 3, k1 = e1 & 7;
    ft->values[(ny0 * 4 + nx0) * 8 + k0] = (uint8_t)roundf(255.0f * v0 / sq);
    ft->values[(ny1 * 4 + nx1) * 8 + k1] = (uint8_t)roundf(255.0f * v1 / sq);
  }
  if (lane == 0) {
    ft->theta = kp.theta;
    ft->sigma = kp.sigma;
    ft->loc.x = kp.loc.x * pixelWidth;
    ft->loc.y = kp.loc.y * pixelWidth;
  }
}
// thrust::remove / remove_if in place, order kept, one pass: a tile's kept elements land at or below where they were read,
// inside the regions of tiles that loaded their elements before they published the aggregate this tile waited for
template <typename T, typename Drop>
__global__ __launch_bounds__(svs::kThreads) void k_x_remove_if(T* data, uint32_t n, Drop drop, uint32_t* __restrict__ count, svs::TileScan<1> ts) {
  constexpr int ITEMS = 4;
  constexpr uint32_t kTile = svs::kThreads * ITEMS;
  for (uint32_t tile = svs::next_tile(ts.counter); tile < ts.numTiles; tile = svs::next_tile(ts.counter)) {
    const uint32_t base = tile * kTile + threadIdx.x * ITEMS;
    T v[ITEMS];
    bool keep[ITEMS];
    uint32_t mine[1] = {0};
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      keep[j] = false;
      if (base + j < n) {
        v[j] = data[base + j];
        keep[j] = !drop(v[j]);
      }
      mine[0] += keep[j] ? 1u : 0u;
    }
    uint32_t excl[1], total[1], prefix[1];
    svs::block_exclusive<1>(mine, excl, total);  // (its barriers: every element of the tile is in registers before a store)
    svs::tile_prefix<1>(ts, tile, total, prefix);
    uint32_t at = prefix[0] + excl[0];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
      if (keep[j]) data[at++] = v[j];
    if (tile == ts.numTiles - 1 && threadIdx.x == 0) *count = prefix[0] + total[0];
  }
}
struct DropMinusOne { __device__ bool operator()(int v) const { return v == -1; } };
struct DropFlagged { __device__ bool operator()(const ssrlcv_sskeypoint& k) const { return k.discard != 0; } };
struct DropNegMax { __device__ bool operator()(float v) const { return v == -FLT_MAX; } };
template <typename T, typename Drop>
int x_remove_if(T* data, uint32_t n, uint32_t* count_dev, void* workspace, size_t workspaceBytes, hipStream_t st) {
  if (!count_dev) return SSRLCV_ERR_INVALID_ARG;
  if (n == 0) return (int)hipMemsetAsync(count_dev, 0, 4, st);
  if (!data || !workspace) return SSRLCV_ERR_INVALID_ARG;
  const uint32_t tiles = (n + svs::kThreads * 4 - 1) / (svs::kThreads * 4);
  if (workspaceBytes < svs::workspace_bytes<1>(tiles)) return SSRLCV_ERR_WORKSPACE;
  SSRLCV_HIP_TRY(hipMemsetAsync(workspace, 0, svs::workspace_bytes<1>(tiles), st));
  hipLaunchKernelGGL((k_x_remove_if<T, Drop>), dim3(tiles < 2048u ? tiles : 2048u), dim3(svs::kThreads), 0, st, data, n, Drop(), count_dev,
                     svs::make_tile_scan<1>(workspace, tiles));
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}
inline unsigned xb(size_t n) { return (unsigned)((n + 255) / 256); }
}  // namespace
extern "C" {

int ssrlcv_hip_find_extrema(uint32_t w, uint32_t h, const float* pixelsUpper, const float* pixelsMiddle, const float* pixelsLower,
                            int* extrema, ssrlcv_stream_t stream) {
  if (!pixelsUpper || !pixelsMiddle || !pixelsLower || !extrema || w < 3 || h < 3) return SSRLCV_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_x_find_extrema, dim3(xb((size_t)w * h)), dim3(256), 0, (hipStream_t)stream, (int)w, (int)h, pixelsUpper, pixelsMiddle,
                     pixelsLower, extrema);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}
size_t ssrlcv_hip_compact_workspace_bytes(uint32_t n) { return svs::workspace_bytes<1>((n + svs::kThreads * 4 - 1) / (svs::kThreads * 4) + 1); }
int ssrlcv_hip_compact_addresses(int* addresses, uint32_t n, uint32_t* count_dev, void* workspace, size_t workspaceBytes,
                                 ssrlcv_stream_t stream) {
  return x_remove_if<int, DropMinusOne>(addresses, n, count_dev, workspace, workspaceBytes, (hipStream_t)stream);
}
int ssrlcv_hip_compact_thetas(float* thetas, uint32_t n, uint32_t* count_dev, void* workspace, size_t workspaceBytes, ssrlcv_stream_t stream) {
  return x_remove_if<float, DropNegMax>(thetas, n, count_dev, workspace, workspaceBytes, (hipStream_t)stream);
}
int ssrlcv_hip_compact_keypoints(ssrlcv_sskeypoint* keyPoints, uint32_t n, uint32_t* count_dev, void* workspace, size_t workspaceBytes,
                                 ssrlcv_stream_t stream) {
  return x_remove_if<ssrlcv_sskeypoint, DropFlagged>(keyPoints, n, count_dev, workspace, workspaceBytes, (hipStream_t)stream);
}
int ssrlcv_hip_fill_extrema(uint32_t numKeyPoints, uint32_t w, uint32_t h, int octave, int blur, float sigma, const int* extremaAddresses,
                            const float* pixels, ssrlcv_sskeypoint* keyPoints, ssrlcv_stream_t stream) {
  (void)h;
  if (numKeyPoints == 0) return SSRLCV_OK;
  if (!extremaAddresses || !pixels || !keyPoints || w == 0) return SSRLCV_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_x_fill_extrema, dim3(xb(numKeyPoints)), dim3(256), 0, (hipStream_t)stream, (int)numKeyPoints, (int)w, octave, blur, sigma,
                     extremaAddresses, pixels, keyPoints);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}
int ssrlcv_hip_flag_noise(uint32_t numKeyPoints, ssrlcv_sskeypoint* keyPoints, float threshold, ssrlcv_stream_t stream) {
  if (numKeyPoints == 0) return SSRLCV_OK;
  if (!keyPoints) return SSRLCV_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_x_flag_noise, dim3(xb(numKeyPoints)), dim3(256), 0, (hipStream_t)stream, numKeyPoints, keyPoints, threshold);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}
int ssrlcv_hip_refine_location(uint32_t numKeyPoints, uint32_t w, uint32_t h, float sigmaMin, float blurSigmaMultiplier, uint32_t numBlurs,
                               const float* const* pixels_dev, ssrlcv_sskeypoint* keyPoints, ssrlcv_stream_t stream) {
  if (numKeyPoints == 0) return SSRLCV_OK;
  if (!pixels_dev || !keyPoints || w < 3 || h < 3 || numBlurs < 3) return SSRLCV_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_x_refine, dim3(xb(numKeyPoints)), dim3(256), 0, (hipStream_t)stream, numKeyPoints, (int)w, (int)h, sigmaMin,
                     blurSigmaMultiplier, (int)numBlurs, MatSource{pixels_dev}, keyPoints);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}
int ssrlcv_hip_flag_edges(uint32_t numKeyPoints, uint32_t startingIndex, uint32_t w, uint32_t h, ssrlcv_sskeypoint* keyPoints,
                          const float* pixels, float threshold, ssrlcv_stream_t stream) {
  (void)h;
  if (numKeyPoints == 0) return SSRLCV_OK;
  if (!keyPoints || !pixels || w < 3) return SSRLCV_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_x_flag_edges, dim3(xb(numKeyPoints)), dim3(256), 0, (hipStream_t)stream, numKeyPoints, startingIndex, (int)w, keyPoints,
                     pixels, threshold);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}
int ssrlcv_hip_check_keypoints(uint32_t numKeyPoints, uint32_t keyPointIndex, uint32_t w, uint32_t h, float pixelWidth, float lambda,
                               ssrlcv_sskeypoint* keyPoints, ssrlcv_stream_t stream) {
  if (numKeyPoints == 0) return SSRLCV_OK;
  if (!keyPoints) return SSRLCV_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_x_check, dim3(xb(numKeyPoints)), dim3(256), 0, (hipStream_t)stream, numKeyPoints, keyPointIndex, (int)w, (int)h, pixelWidth,
                     lambda, keyPoints);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}
int ssrlcv_hip_pixel_gradients(uint32_t w, uint32_t h, const float* pixels, ssrlcv_float2* gradients, ssrlcv_stream_t stream) {
  if (!pixels || !gradients || w < 2 || h < 2) return SSRLCV_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_x_gradients, dim3(xb((size_t)w * h)), dim3(256), 0, (hipStream_t)stream, (int)w, (int)h, pixels, (float2*)gradients);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}
int ssrlcv_hip_compute_thetas(uint32_t numKeyPoints, uint32_t keyPointIndex, uint32_t w, uint32_t h, float pixelWidth, float lambda,
                              const ssrlcv_sskeypoint* keyPoints, const ssrlcv_float2* gradients, int* thetaNumbers, uint32_t maxOrientations,
                              float orientationThreshold, float* thetas, ssrlcv_stream_t stream) {
  if (numKeyPoints == 0) return SSRLCV_OK;
  if (!keyPoints || !gradients || !thetaNumbers || !thetas || maxOrientations == 0 || maxOrientations > 8) return SSRLCV_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_x_thetas, dim3((numKeyPoints + 63) / 64), dim3(64), 0, (hipStream_t)stream, numKeyPoints, keyPointIndex, (int)w, (int)h,
                     pixelWidth, lambda, keyPoints, (const float2*)gradients, thetaNumbers, maxOrientations, orientationThreshold, thetas);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}
int ssrlcv_hip_expand_keypoints(uint32_t numKeyPoints, const ssrlcv_sskeypoint* keyPointsIn, ssrlcv_sskeypoint* keyPointsOut,
                                const int* thetaAddresses, const float* thetas, ssrlcv_stream_t stream) {
  if (numKeyPoints == 0) return SSRLCV_OK;
  if (!keyPointsIn || !keyPointsOut || !thetaAddresses || !thetas) return SSRLCV_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_x_expand, dim3(xb(numKeyPoints)), dim3(256), 0, (hipStream_t)stream, numKeyPoints, keyPointsIn, keyPointsOut, thetaAddresses,
                     thetas);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}
int ssrlcv_hip_fill_descriptors(uint32_t numFeatures, uint32_t keyPointIndex, uint32_t w, uint32_t h, ssrlcv_sift_feature* features,
                                float pixelWidth, float lambda, const ssrlcv_sskeypoint* keyPoints, const ssrlcv_float2* gradients,
                                ssrlcv_stream_t stream) {
  if (numFeatures == 0) return SSRLCV_OK;
  if (!features || !keyPoints || !gradients) return SSRLCV_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_x_descriptors, dim3(numFeatures), dim3(64), 0, (hipStream_t)stream, numFeatures, keyPointIndex, (int)w, (int)h, pixelWidth,
                     lambda, keyPoints, (const float2*)gradients, features);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

int ssrlcv_hip_sift_extract(const ssrlcv_sift_plan* plan, const uint8_t* pixels, void* workspace,
                            ssrlcv_sift_feature* features, uint32_t* numFeatures, ssrlcv_stream_t stream) {
  if (!plan) return SSRLCV_ERR_INVALID_ARG;
  plan->fusedCall = 1;  // build_dog may leave the polar tables in flight: describe joins them below
  int rc = ssrlcv_hip_sift_build_dog(plan, pixels, workspace, stream);
  plan->fusedCall = 0;
  if (rc) {  // build_dog may already have queued octave 0's list chain (or the tables) on a side stream
    drain_side_streams(plan, (hipStream_t)stream);
    return rc;
  }
  if (plan->stageEvent && hipEventRecord(plan->stageEvent, (hipStream_t)stream) != hipSuccess) return (int)hipGetLastError();
  return ssrlcv_hip_sift_describe(plan, workspace, features, numFeatures, stream);
}

}  // extern "C"
