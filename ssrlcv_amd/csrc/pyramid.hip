// ssrlcv_amd/csrc/pyramid.hip -- DoG scale-space pyramid for gfx950.
//
// The pyramid is HBM-bound stencil work.  Design choices (DESIGN.md "Pyramid"):
//   * separable Gaussian = ONE kernel per level: 256-column strips march down the image 8 rows at a time.  Input rows
//     are staged in LDS with their horizontal halo (coalesced, mirrored border), the horizontal pass runs from LDS
//     with an 8-outputs-per-thread register block, the vertical pass keeps a sliding window of 2R+8 horizontally
//     filtered rows IN REGISTERS (one column per thread), so every input pixel is fetched once per strip and every
//     tap costs one v_fma_f32 whose weight operand is an SGPR (weights travel in the kernarg segment); from 23 taps
//     up the same two passes run as banded-Toeplitz products on the f32 matrix cores (k_gauss_mfma), bit-identical;
//   * accumulation order is the reference's (k = -R..R, fmaf chain), so levels are bit-identical to the oracle;
//   * the level's global min/max (normalizeImage finds it on the host after a D2H copy, src/Image.cu:631-649) is
//     reduced in the epilogue of the same kernel: block reduction + one integer atomic pair per block;
//   * both normalisations + the DoG subtraction are one streaming kernel reading 6 levels and writing 5 (float4
//     lanes), which also reduces the DoG levels' min/max for findKeyPoints' second normalisation.
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include <float.h>
#include <math.h>
#include <new>
#include <stdlib.h>
#include <string.h>
#include "dev_switch.h"
#include "device_math.h"
#include "sift_plan.h"
#include "ssrlcv_hip.h"

namespace {

// src/Image.cu:1248-1252 getSymmetrizedCoord
__device__ __forceinline__ int sym_coord(int i, int l) {
  int ll = 2 * l;
  if (__builtin_expect(i < -ll || i >= ll, 0)) i = (i + ll) % ll;  // only tiny images reach this
  if (i < 0) i += ll;        // (i + ll) % ll for -ll <= i < 0
  return (i > l - 1) ? ll - 1 - i : i;
}

// Mirror of the one-tile kernel: coordinates beyond the last pixel a valid output needs only meet zero weights, so they are
// clamped first and one reflection is enough for sides of at least 2R (every level of a 4096^2 image); smaller levels
// take the reference's modulo (sides down to 16 pixels: below that upstream itself reads out of bounds, see plan_create).
__device__ __forceinline__ int mirror_into(int i, int l, int R) {
  if (l < 64) return sym_coord(i, l);
  i = i > l - 1 + R ? l - 1 + R : i;
  i = i < 0 ? -1 - i : i;
  return i > l - 1 ? 2 * l - 1 - i : i;
}

// float atomics on {min,max} via the integer-ordering trick (same idea as atomicMinFloat/atomicMaxFloat,
// src/FeatureFactory.cu:769-780)
__device__ __forceinline__ void atomic_min_f(float* addr, float v) {
  if (v >= 0) atomicMin((int*)addr, __float_as_int(v));
  else atomicMax((unsigned int*)addr, __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_f(float* addr, float v) {
  if (v >= 0) atomicMax((int*)addr, __float_as_int(v));
  else atomicMin((unsigned int*)addr, __float_as_uint(v));
}
__device__ __forceinline__ void wave_minmax_commit(float mn, float mx, float* minmax) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o, 64));
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomic_min_f(minmax, mn);
    atomic_max_f(minmax + 1, mx);
  }
}

// Block-level variant: one atomic pair per BLOCK.  Same-address float atomics serialise at the memory side (~12-50 ns
// each), so one pair per wave cost 1.5 ms on a 67 Mpx DoG launch; per block it is noise.  Every thread of the block
// must call it (it contains __syncthreads).
__device__ __forceinline__ void block_minmax_commit(float mn, float mx, float* minmax, float* s_red /* >= 8 floats */) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o, 64));
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  }
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { s_red[2 * wave] = mn; s_red[2 * wave + 1] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < nw; ++w) { mn = fminf(mn, s_red[2 * w]); mx = fmaxf(mx, s_red[2 * w + 1]); }
    atomic_min_f(minmax, mn);
    atomic_max_f(minmax + 1, mx);
  }
}

// generateBW (src/Image.cu:1277-1296) with bwaToBW / rgbToBW / rgbaToBW (:1253-1261): integer arithmetic on promoted
// bytes.  (1 - a) * x + a * x is x for every alpha, so 2 channels keep the grey value, 4 channels drop alpha and go
// through the RGB rule r/4 + g/2 + b/4 (each quotient truncated, sum <= 253).
__global__ __launch_bounds__(256) void k_to_bw(const uint8_t* __restrict__ in, int depth, uint8_t* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint8_t* p = in + i * (size_t)depth;
  out[i] = depth == 2 ? p[0] : (uint8_t)((p[0] / 4) + (p[1] / 2) + (p[2] / 4));
}

__global__ void k_init_minmax(float* mm, int pairs) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < pairs) { mm[2 * i] = FLT_MAX; mm[2 * i + 1] = -FLT_MAX; }
}

// ---- S1 / S2 / S5 ------------------------------------------------------------------------------------------------
// S3: addBufferBorder (src/Image.cu:572-598): the image inside a zero border of (bx, by) pixels; grid = padded size
template <typename T>
__global__ __launch_bounds__(256) void k_add_border(const T* __restrict__ in, uint32_t w, uint32_t h, T* __restrict__ out,
                                                    uint32_t bx, uint32_t by) {
  const uint32_t ow = w + 2 * bx, x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= ow) return;
  const uint32_t sx = x - bx, sy = y - by;  // wraps to a huge value left of / above the image
  out[(size_t)y * ow + x] = (sx < w && sy < h) ? in[(size_t)sy * w + sx] : (T)0;
}

__global__ __launch_bounds__(256) void k_u8_to_f32(const uint8_t* __restrict__ in, float* __restrict__ out, size_t n) {
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    uchar4 v = *reinterpret_cast<const uchar4*>(in + i);
    float4 f = make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
    *reinterpret_cast<float4*>(out + i) = f;
  } else {
    for (; i < n; ++i) out[i] = (float)in[i];
  }
}

// upsampleImage(float) (src/Image.cu:1393-1414); T = float or uint8_t source (the u8 variant fuses convertToFltImage)
template <typename T>
__device__ __forceinline__ float upsample_at(const T* __restrict__ in, int w, int h, int i, int j) {
  float x = i * 0.5f;
  float y = j * 0.5f;
  int xm = sym_coord((int)x, w);
  int xp = sym_coord((int)x + 1, w);
  int ym = sym_coord((int)y, h);
  int yp = sym_coord((int)y + 1, h);
  float dx = x - floorf(x), dy = y - floorf(y);
  float sumPix = dx * dy * ((float)in[(size_t)yp * w + xp]);
  sumPix += (1.0f - dx) * dy * ((float)in[(size_t)yp * w + xm]);
  sumPix += dx * (1 - dy) * ((float)in[(size_t)ym * w + xp]);
  sumPix += (1 - dx) * (1 - dy) * ((float)in[(size_t)ym * w + xm]);
  return sumPix;
}
template <typename T>
__global__ __launch_bounds__(256) void k_upsample2x(const T* __restrict__ in, uint32_t w, uint32_t h,
                                                    float* __restrict__ out) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  uint32_t j = blockIdx.y;
  if (i < w * 2 && j < h * 2) out[(size_t)j * (w * 2) + i] = upsample_at(in, (int)w, (int)h, (int)i, (int)j);
}

// u8 source, four outputs per thread.  With 8-bit pixels and bilinear weights in {0, 1/4, 1/2, 1} every product and
// partial sum of upsample_at is exact in fp32, so its result equals (sum of the four taps) / 4 evaluated in integers:
// out(i, j) = 0.25 * (p[ym][xm] + p[ym][xp'] + p[yp'][xm] + p[yp'][xp']) with xp' = xp when i is odd, else xm (same in y).
__global__ __launch_bounds__(256) void k_upsample2x_u8x4(const uint8_t* __restrict__ in, uint32_t w, uint32_t h,
                                                         float* __restrict__ out) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;  // outputs 4t .. 4t+3 of row j
  const uint32_t j = blockIdx.y;
  if (4 * t >= 2 * w) return;
  const uint32_t c0 = 2 * t, c1 = 2 * t + 1, c2 = (2 * t + 2 < w) ? 2 * t + 2 : w - 1;
  const uint32_t ym = j >> 1, yp = (j & 1) ? ((ym + 1 < h) ? ym + 1 : h - 1) : ym;
  const uint8_t* r0 = in + (size_t)ym * w;
  const uint8_t* r1 = in + (size_t)yp * w;
  const uint32_t a0 = r0[c0] + (uint32_t)r1[c0], a1 = r0[c1] + (uint32_t)r1[c1], a2 = r0[c2] + (uint32_t)r1[c2];
  float4 o;
  o.x = (float)(2 * a0) * 0.25f;
  o.y = (float)(a0 + a1) * 0.25f;
  o.z = (float)(2 * a1) * 0.25f;
  o.w = (float)(a1 + a2) * 0.25f;
  *reinterpret_cast<float4*>(out + (size_t)j * (2 * w) + 4 * t) = o;
}

// binImage(float) (src/Image.cu:1380-1392)
__global__ __launch_bounds__(256) void k_bin2x(const float* __restrict__ in, uint32_t w, uint32_t h,
                                               float* __restrict__ out) {
  uint32_t x = blockIdx.x * 256 + threadIdx.x;
  uint32_t y = blockIdx.y;
  uint32_t ow = w / 2;
  if (x < ow && y < h / 2) {
    float2 r0 = *reinterpret_cast<const float2*>(in + (size_t)(y * 2) * w + x * 2);
    float2 r1 = *reinterpret_cast<const float2*>(in + (size_t)(y * 2 + 1) * w + x * 2);
    float sumPix = r0.x + r1.x + r0.y + r1.y;  // reference operand order
    out[(size_t)y * ow + x] = sumPix / 4.0f;
  }
}

// ---- S4: fused separable Gaussian --------------------------------------------------------------------------------
struct ConvArgs {
  const float* in;
  float* out;
  float* minmax;  // nullable
  uint32_t w, h;
  uint32_t rowsPerBlock;
  uint32_t x0base;  // first column of the launch's first strip (k_gauss_fused on the partial last strip)
  const uint8_t* u8;  // k_gauss_strip<R, true>: the u8 image (w/2 x h/2) whose 2x bilinear upsample is the input
  // k_gauss_mfma2 / k_gauss_tile: also emit the 2x2 bin of the output (S5, the next octave's input) from the accumulators
  float* binOut;  // (w/2) x (h/2), nullptr: no bin
  float wgt[33];  // taps are symmetric (w[k] == w[2R-k] bit for bit): only k = 0..R travel, in SGPRs
  uint32_t xcdStrips;  // strip kernels: neighbouring strips on the same XCD (strip_block)
#ifdef SSRLCV_STAMPS
  long long* stamps;  // tools/gauss_lab.hip: s_memtime stamps of one block, [wave][step][8]
#endif
};
// Which strip (x) and row block (y) a workgroup of a strip kernel takes.  Workgroups go to the eight XCDs round-robin by their
// linear id, so with the plain mapping the strips left and right of a strip -- which read the same 2R halo columns -- run on
// other XCDs and every halo column is pulled through two L2s.  With a strip count that is a multiple of eight each XCD takes
// gridDim.x / 8 neighbouring strips, all of their row blocks, in the launch order of its workgroups (row block by row block,
// so that neighbours march down the image together).
__device__ __forceinline__ void strip_block(bool xcdStrips, uint32_t& sx, uint32_t& by) {
  sx = blockIdx.x;
  by = blockIdx.y;
  const uint32_t nx = gridDim.x;
  if (xcdStrips && (nx & 7u) == 0u) {
    const uint32_t p = by * nx + sx, per = nx >> 3, q = p >> 3;
    sx = (p & 7u) * per + q % per;
    by = q / per;
  }
}

// Instrumentation (results unchanged): -DSSRLCV_STAMPS builds write s_memtime stamps of one block into a buffer of their own.
// Only the tools/ lab programs build that way (csrc/dev_switch.h refuses the define in a library build).
#ifdef SSRLCV_STAMPS
long long* g_lab_stamps = nullptr;
#define SV_STAMP(slot)                                                                                   \
  do {                                                                                                    \
    if (a.stamps && blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && it < 64) {              \
      const long long t_ = (long long)__builtin_amdgcn_s_memtime();                                       \
      if (lane == 0) a.stamps[((size_t)wave * 64 + it) * 8 + (slot)] = t_;                                \
    }                                                                                                     \
  } while (0)
#else
#define SV_STAMP(slot) do { } while (0)
#endif

constexpr int kTX = 256;  // strip width = threads per block
constexpr int kNR = 8;    // rows per marching step

template <int R>
__global__ __launch_bounds__(kTX) void k_gauss_fused(ConvArgs a) {
  constexpr int RP = (R + 3) / 4 * 4;   // halo rounded up so LDS reads stay 16-byte aligned
  constexpr int SW = kTX + 2 * RP;      // staged row width (floats)
  constexpr int WIN = 2 * R + kNR;      // vertical register window
  constexpr int HQ = (2 * RP + 8) / 4;  // float4 reads of the horizontal pass
  constexpr int TOT = kNR * SW;         // staged floats per step
  constexpr int STG = (TOT + kTX - 1) / kTX;
  __shared__ float4 s_in4[kNR][SW / 4];
  __shared__ float4 s_h4[kNR][kTX / 4];
  float* s_in = reinterpret_cast<float*>(&s_in4[0][0]);
  float* s_h = reinterpret_cast<float*>(&s_h4[0][0]);

  const int W = (int)a.w, H = (int)a.h;
  const int x0 = (int)a.x0base + blockIdx.x * kTX;
  const int y0 = blockIdx.y * (int)a.rowsPerBlock;
  int nrows = (int)a.rowsPerBlock;
  if (y0 + nrows > H) nrows = H - y0;
  const int tid = threadIdx.x;
  const int hr = tid >> 5, hq = (tid & 31) * 2;  // horizontal-pass role: row hr, outputs 8*(tid&31) .. +7

  // staging role: element e of this thread is float (e*256 + tid) of the [kNR][SW] tile; its column never changes.
  // Columns/rows beyond the last pixel any valid output needs are clamped first, so the mirror needs no modulo
  // (host guarantees W,H >= 64 > R).
  int gxs[STG], rws[STG];
#pragma unroll
  for (int e = 0; e < STG; ++e) {
    int idx = e * kTX + tid;
    int r = idx / SW, c = idx - r * SW;
    int x = x0 - RP + c;
    x = x > W - 1 + R ? W - 1 + R : x;
    x = x < 0 ? -1 - x : x;                 // sym_coord for -l <= i < 0
    x = x > W - 1 ? 2 * W - 1 - x : x;      // sym_coord for l <= i < 2l
    gxs[e] = x;
    rws[e] = r;
  }
  float win[WIN];
#pragma unroll
  for (int k = 0; k < WIN; ++k) win[k] = 0.0f;
  float mn = FLT_MAX, mx = -FLT_MAX;
  const int steps = (nrows + 2 * R + kNR - 1) / kNR;
  float pre[STG];
  auto fetch = [&](int s) {
    const int ybase = y0 - R + s * kNR;
#pragma unroll
    for (int e = 0; e < STG; ++e) {
      int y = ybase + rws[e];
      y = y > H - 1 + R ? H - 1 + R : y;
      y = y < 0 ? -1 - y : y;
      y = y > H - 1 ? 2 * H - 1 - y : y;
      if ((e + 1) * kTX <= TOT || e * kTX + tid < TOT) pre[e] = a.in[(size_t)y * W + gxs[e]];
    }
  };
  fetch(0);
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int e = 0; e < STG; ++e)
      if ((e + 1) * kTX <= TOT || e * kTX + tid < TOT) s_in[e * kTX + tid] = pre[e];
    __syncthreads();
    if (s + 1 < steps) fetch(s + 1);  // global loads stay in flight under the two passes below
    // horizontal pass: out[i] = sum_k w[k] * row[8*(tid&31) + i + k - R]; the window streams through 4 registers at a
    // time and every output's fmaf chain still runs k = 0..2R in order
    {
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = 0.0f;
#pragma unroll
      for (int q = 0; q < HQ; ++q) {
        float4 v4 = s_in4[hr][hq + q];
        float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int k = 4 * q + j - (RP - R) - i;
            if (k >= 0 && k <= 2 * R) o[i] = __builtin_fmaf(v[j], a.wgt[k <= R ? k : 2 * R - k], o[i]);
          }
        }
      }
      s_h4[hr][hq] = make_float4(o[0], o[1], o[2], o[3]);
      s_h4[hr][hq + 1] = make_float4(o[4], o[5], o[6], o[7]);
    }
    __syncthreads();
    // vertical pass: this thread owns column x0 + tid; sliding window of 2R + 8 filtered rows in registers
#pragma unroll
    for (int i = 0; i < kNR; ++i) win[2 * R + i] = s_h[i * kTX + tid];
    const int gx = x0 + tid;
    if (s * kNR + kNR - 1 >= 2 * R) {  // the first 2R / kNR steps only fill the window: every output row is < 0
      // k outer / i inner: eight independent fmaf chains in flight (each output's own chain still runs k = 0..2R)
      float vs[kNR];
#pragma unroll
      for (int i = 0; i < kNR; ++i) vs[i] = 0.0f;
#pragma unroll
      for (int k = 0; k <= 2 * R; ++k) {
#pragma unroll
        for (int i = 0; i < kNR; ++i) vs[i] = __builtin_fmaf(win[i + k], a.wgt[k <= R ? k : 2 * R - k], vs[i]);
      }
#pragma unroll
      for (int i = 0; i < kNR; ++i) {
        int j = s * kNR - 2 * R + i;
        if (j >= 0 && j < nrows && gx < W) {
          a.out[(size_t)(y0 + j) * W + gx] = vs[i];
          mn = fminf(mn, vs[i]);
          mx = fmaxf(mx, vs[i]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 2 * R; ++k) win[k] = win[k + kNR];
  }
  if (a.minmax) block_minmax_commit(mn, mx, a.minmax, s_h);
}

// The same kernel for strips that lie fully inside the image (x0 + 256 <= W, rows 16-byte aligned): the per-element
// mirror / address arithmetic of the generic staging above was as many VALU instructions as the 16 (2R + 1) FMAs of a
// step.  Here a wave stages two whole rows per step: the row index is wave-uniform (mirrored in SGPRs), every lane loads
// one float4 of the strip's interior and lanes 0 .. 2 RP - 1 one halo float each (columns mirrored once, outside the
// loop); stores go to a uniform row base.  Passes and summation order are identical, so are the results.
// UPS = true: the input is the 2x bilinear upsample of a u8 image (S1 + S2), formed in the loader instead of being
// written out by k_upsample2x_u8x4 and read back: level 0 of octave 0 is HBM bound (268 MB in + 268 MB out per 4096^2
// image) and then reads 17 MB.  The value of an upsampled pixel is the exact integer form of upsample_at (see
// k_upsample2x_u8x4): 0.25 * (p[ym][xm] + p[ym][xp'] + p[yp'][xm] + p[yp'][xp']), with the mirror of the convolution
// applied to the upsampled coordinates first.
#ifndef SSRLCV_STRIP_VGPR_WEIGHTS
#define SSRLCV_STRIP_VGPR_WEIGHTS 0  // measured in round 5: 178 -> 182 us per image over the 17-tap launches (profiles/r05_kernel_ab.txt)
#endif
template <int R, bool UPS>
__global__ __launch_bounds__(kTX) void k_gauss_strip(ConvArgs a) {
  constexpr int RP = (R + 3) / 4 * 4;
  constexpr int SW = kTX + 2 * RP;
  constexpr int WIN = 2 * R + kNR;
  constexpr int HQ = (2 * RP + 8) / 4;
  static_assert(kNR == 8 && kTX == 256, "two rows per wave and step");
  __shared__ float4 s_in4[kNR][SW / 4];
  __shared__ float4 s_h4[kNR][kTX / 4];
  float* s_in = reinterpret_cast<float*>(&s_in4[0][0]);
  float* s_h = reinterpret_cast<float*>(&s_h4[0][0]);

  const int W = (int)a.w, H = (int)a.h;
  uint32_t sbx, sby;
  strip_block(a.xcdStrips != 0u, sbx, sby);
  const int x0 = (int)sbx * kTX;  // host: x0 + kTX <= W
  const int y0 = (int)sby * (int)a.rowsPerBlock;
  int nrows = (int)a.rowsPerBlock;
  if (y0 + nrows > H) nrows = H - y0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hr = tid >> 5, hq = (tid & 31) * 2;  // horizontal-pass role: row hr, outputs 8*(tid&31) .. +7

  // halo role: lane h < RP owns column x0 - RP + h, lane RP <= h < 2 RP column x0 + 256 + (h - RP)
  const bool halo = lane < 2 * RP;
  int hx = lane < RP ? x0 - RP + lane : x0 + kTX + (lane - RP);
  hx = hx > W - 1 + R ? W - 1 + R : hx;
  hx = hx < 0 ? -1 - hx : hx;              // sym_coord for -l <= i < 0
  hx = hx > W - 1 ? 2 * W - 1 - hx : hx;   // sym_coord for l <= i < 2l
  const int hl = lane < RP ? lane : kTX + lane;  // its column in the staged row: the right halo starts at RP + 256

  float win[WIN];
#pragma unroll
  for (int k = 0; k < WIN; ++k) win[k] = 0.0f;
  // SSRLCV_STRIP_VGPR_WEIGHTS=1 keeps the R + 1 distinct weights in VECTOR registers (an FMA with a scalar source issues in
  // ~4.2 cycles per SIMD against ~3.3 with three vector sources, tools/valu_rate.hip): no faster, the kernel is not bound
  // by FMA issue; the default leaves them in SGPRs.
  float wv[R + 1];
#pragma unroll
  for (int k = 0; k <= R; ++k) {
    wv[k] = a.wgt[k];
#if SSRLCV_STRIP_VGPR_WEIGHTS
    asm volatile("" : "+v"(wv[k]));
#endif
  }
  float mn = FLT_MAX, mx = -FLT_MAX;
  const int steps = (nrows + 2 * R + kNR - 1) / kNR;
  float4 preI0, preI1;  // next step's two rows of this wave (named scalars: an indexed pair went to scratch)
  float preH0 = 0.0f, preH1 = 0.0f;
  auto mirror_row = [&](int y) {  // wave-uniform
    y = y > H - 1 + R ? H - 1 + R : y;
    y = y < 0 ? -1 - y : y;
    y = y > H - 1 ? 2 * H - 1 - y : y;
    return y;
  };
  auto row_of = [&](int y) { return a.in + (size_t)mirror_row(y) * W; };
  // UPS: one upsampled row Y of this lane's four columns and of its halo column, from u8 rows ym = Y >> 1 and yp'
  const int uw = W >> 1, uh = H >> 1;
  const int c0 = (x0 >> 1) + 2 * lane;                  // source column of outputs 4 lane, 4 lane + 1 (even: 2-byte aligned)
  const int c2 = c0 + 2 < uw ? c0 + 2 : uw - 1;
  const int hxm = hx >> 1, hxp = (hx & 1) ? (hxm + 1 < uw ? hxm + 1 : uw - 1) : hxm;
  // The u8 loads of a row and their conversion are two steps: the loads are issued behind the barrier and fly under the
  // two passes, the integer sums and conversions run at the end of the step (converted at once, the wave waited for its
  // byte loads in front of the horizontal pass: 55 % of its cycles in s_waitcnt, SQ_WAIT_ANY).
  struct RawRow { uint32_t p0, p1, q0, q1, h00, h01, h10, h11; };
  RawRow rawA = {0, 0, 0, 0, 0, 0, 0, 0}, rawB = {0, 0, 0, 0, 0, 0, 0, 0};
  auto up_load = [&](int Y, RawRow& r) {
    Y = mirror_row(Y);
    const int ym = Y >> 1, yp = (Y & 1) ? (ym + 1 < uh ? ym + 1 : uh - 1) : ym;
    const uint8_t* r0 = a.u8 + (size_t)ym * uw;
    const uint8_t* r1 = a.u8 + (size_t)yp * uw;
    r.p0 = *reinterpret_cast<const uint16_t*>(r0 + c0);
    r.p1 = *reinterpret_cast<const uint16_t*>(r1 + c0);
    r.q0 = r0[c2];
    r.q1 = r1[c2];
    if (halo) { r.h00 = r0[hxm]; r.h01 = r0[hxp]; r.h10 = r1[hxm]; r.h11 = r1[hxp]; }
  };
  auto up_convert = [&](const RawRow& r, float4& o, float& oh) {
    const uint32_t a0 = (r.p0 & 255u) + (r.p1 & 255u), a1 = (r.p0 >> 8) + (r.p1 >> 8), a2 = r.q0 + r.q1;
    o.x = (float)(2 * a0) * 0.25f;
    o.y = (float)(a0 + a1) * 0.25f;
    o.z = (float)(2 * a1) * 0.25f;
    o.w = (float)(a1 + a2) * 0.25f;
    if (halo) oh = (float)(r.h00 + r.h01 + r.h10 + r.h11) * 0.25f;
  };
  auto fetch = [&](int s) {
    const int ybase = y0 - R + s * kNR + 2 * wave;
    if (UPS) {
      up_load(ybase, rawA);
      up_load(ybase + 1, rawB);
      return;
    }
    const float* r0 = row_of(ybase);
    const float* r1 = row_of(ybase + 1);
    preI0 = *reinterpret_cast<const float4*>(r0 + x0 + 4 * lane);
    preI1 = *reinterpret_cast<const float4*>(r1 + x0 + 4 * lane);
    if (halo) {
      preH0 = r0[hx];
      preH1 = r1[hx];
    }
  };
  auto convert = [&]() {
    if (UPS) {
      up_convert(rawA, preI0, preH0);
      up_convert(rawB, preI1, preH1);
    }
  };
  fetch(0);
  convert();
  const int gx = x0 + tid;
  for (int s = 0; s < steps; ++s) {
    s_in4[2 * wave][RP / 4 + lane] = preI0;
    s_in4[2 * wave + 1][RP / 4 + lane] = preI1;
    if (halo) {
      s_in[(2 * wave) * SW + hl] = preH0;
      s_in[(2 * wave + 1) * SW + hl] = preH1;
    }
    __syncthreads();
    if (s + 1 < steps) fetch(s + 1);  // global loads stay in flight under the two passes below
    {
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = 0.0f;
#pragma unroll
      for (int q = 0; q < HQ; ++q) {
        float4 v4 = s_in4[hr][hq + q];
        float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int k = 4 * q + j - (RP - R) - i;
            if (k >= 0 && k <= 2 * R) o[i] = __builtin_fmaf(v[j], wv[k <= R ? k : 2 * R - k], o[i]);
          }
        }
      }
      s_h4[hr][hq] = make_float4(o[0], o[1], o[2], o[3]);
      s_h4[hr][hq + 1] = make_float4(o[4], o[5], o[6], o[7]);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kNR; ++i) win[2 * R + i] = s_h[i * kTX + tid];
    if (s * kNR + kNR - 1 >= 2 * R) {
      float vs[kNR];
#pragma unroll
      for (int i = 0; i < kNR; ++i) vs[i] = 0.0f;
#pragma unroll
      for (int k = 0; k <= 2 * R; ++k) {
#pragma unroll
        for (int i = 0; i < kNR; ++i) vs[i] = __builtin_fmaf(win[i + k], wv[k <= R ? k : 2 * R - k], vs[i]);
      }
      float* orow = a.out + (size_t)(y0 + s * kNR - 2 * R) * W + gx;
#pragma unroll
      for (int i = 0; i < kNR; ++i) {
        const int j = s * kNR - 2 * R + i;  // uniform
        if (j >= 0 && j < nrows) {
          orow[(size_t)i * W] = vs[i];
          mn = fminf(mn, vs[i]);
          mx = fmaxf(mx, vs[i]);
        }
      }
    }
    if (s + 1 < steps) convert();  // UPS: the bytes requested behind the barrier have arrived under the two passes
#pragma unroll
    for (int k = 0; k < 2 * R; ++k) win[k] = win[k + kNR];
  }
  if (a.minmax) block_minmax_commit(mn, mx, a.minmax, s_h);
}

// ---- S4 on the matrix cores ----------------------------------------------------------------------------------------------
// The two 1-D passes as banded-Toeplitz products on v_mfma_f32_16x16x4_f32.  An f32 MFMA is bit-for-bit a k-ordered
// fmaf chain (one rounding per product-add), so D[i][j] = sum_kk A[i][kk] * T[kk][j] with T[kk][j] = w[kk - j]
// (0 outside the band: fmaf(x, 0, acc) == acc) is exactly the reference's `sum += p * k` loop in its tap order.  The
// f32 MFMA pipe issues 32 FMA per cycle per SIMD against ~14 of the VALU formulation (tools/valu_rate.hip), and the
// band fills 65/80 of a 16-wide tile at R = 32.
//   horizontal: D[16 rows][16 cols] = In[16 rows][16+2R cols] x T      A from the LDS row stage, B = per-lane constants
//   vertical  : D^T[16 cols][16 rows] = H^T[16 cols][16+2R rows] x T             A from the LDS ring of H rows, B = constants
// A block owns a 256- or 128-column strip and marches down 16 rows per step.  It has 8 waves in two roles, one wave of
// each per SIMD: four waves stage the input rows and run the horizontal pass of step `it` (their share of the column
// tiles), the other four run the vertical pass of step `it - 1` from the ring of H rows and store the result.
// (The first edition of this kernel, with per-element staging and the address arithmetic inside the MFMA loops, is in
// the history up to round 2; k_gauss_mfma2 below replaced it, partial last strips go to the VALU kernel.)
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMT = 16;            // rows per marching step of the MFMA kernel
constexpr int kMfmaThreads = 512;  // 4 horizontal-pass waves + 4 vertical-pass waves

template <int R, int TW>
struct MfmaCfg {
  static constexpr int K = 16 + 2 * R;            // Toeplitz depth
  static constexpr int KS = (K + 3) / 4;          // MFMA k-steps
  static constexpr int KP = KS * 4;
  static constexpr int RP = (R + 3) / 4 * 4;
  // staged row width: 256 + halo + padding for the KP - K overshoot, == 2 (mod 32) so that the A reads
  // (16 rows x 2 columns per 32-lane half) are bank-conflict free
  static constexpr int TPW = TW / 64;             // 16-column tiles per wave (4 waves of each role share the strip)
  static constexpr int SWmin = TW + 2 * RP + (KP - K) + 4;
  static constexpr int SW = (SWmin + 29) / 32 * 32 + 2;
  // ring of H rows: the vertical pass of step it-1 reads rows 16(it-1)-2R .. 16 it - 1 while the horizontal pass of
  // step it writes rows 16 it .. 16 it + 15
  static constexpr int RINGROWS = (2 * R + 2 * kMT + kMT - 1) / kMT * kMT;
  static constexpr int RSTR = TW + 16;            // ring row stride (== 16 mod 32: B reads conflict free)
  static constexpr int TOT = kMT * SW;
  static constexpr int STG = (TOT + 255) / 256;   // staged elements per horizontal-role thread (256 of them)
  static constexpr size_t ldsBytes = sizeof(float) * ((size_t)2 * kMT * SW + (size_t)RINGROWS * RSTR);
};

// S5 folded into a vertical-pass epilogue: lane (li, lk) holds four consecutive x of output row y (= row li of a 16-row
// tile that starts on an even row); the row below sits in lane li ^ 1 of the same quad (DPP quad_perm [1,0,3,2]).  Even
// lanes then form binImage's ((r0.x + r1.x) + r0.y + r1.y) / 4 (src/Image.cu:1380-1392) for their two column pairs and
// store them -- the level is not read again by a bin kernel.  `valid`: both rows and all four columns are inside the image.
__device__ __forceinline__ void bin2x_from_tile(const f32x4 v, bool valid, float* __restrict__ binOut, size_t binIndex /* of the first pair */) {
  float p[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    p[r] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (float)v[r]), 0xB1, 0xF, 0xF, true));
  if (valid && (threadIdx.x & 1) == 0) {
    const float s0 = v[0] + p[0] + v[1] + p[1];  // reference operand order (left to right)
    const float s1 = v[2] + p[2] + v[3] + p[3];
    typedef float f32x2b __attribute__((ext_vector_type(2)));
    *reinterpret_cast<f32x2b*>(binOut + binIndex) = f32x2b{s0 / 4.0f, s1 / 4.0f};
  }
}

// ---- k_gauss_mfma, second edition: the same products with (almost) no vector instruction beside them ----------------
// An f32 MFMA runs on the SIMD's fp32 FMA lanes (v_mfma_f32_16x16x4_f32 issues at the vector fp32 rate and
// SQ_VALU_MFMA_COEXEC_CYCLES is 0 on these kernels): while one executes, no vector-ALU instruction of EITHER wave of the
// SIMD issues, and an MFMA behind a vector instruction waits for it -- tools/mfma_lds_rate.hip: ONE v_add_u32 between the
// k-steps of the vertical pass's loop (4 MFMAs) takes the loop from 35.6 to 45 cycles per MFMA, i.e. a lone vector
// instruction among MFMAs costs a whole MFMA slot (~37 cycles), whichever wave it comes from.  s_memtime stamps in
// k_gauss_mfma (tools/gauss_lab.hip, 65 taps, cycles per 16-row step of 8024, 5120 of them MFMA issue): the vertical
// wave's MFMA loop took 4138 instead of 2560 -- 100 vector instructions of ring-offset arithmetic in front of it and
// the horizontal wave's address arithmetic, exec-mask regions and LDS address adds dripping in between its MFMAs.
// This edition keeps the products, their order and the LDS layout and removes the vector instructions:
//  * radii are padded to even values (11 -> 12, 23 -> 24: zero taps, same k-step count), so k-step ks of the vertical
//    pass reads four rows of ONE 16-row ring tile: tile and in-tile row are compile-time, the tile's slot is scalar;
//  * every LDS address of a step is formed at the END of the previous step in ONE batch of v_add (scalar slot / stage
//    offsets + static lane parts) and pinned; inside the loops there are only MFMAs, LDS and scalar instructions;
//  * rows are fetched through a buffer descriptor (row = scalar byte offset, lane part static; the mirror only on the
//    steps that touch the image border), the four halo accesses share one exec region;
//  * the result is stored through a scalar row base + static lane offset, unmasked on interior steps; min / max are one
//    batch of v_min3 / v_max3.
// Only for full, aligned strips (the row-staged case); everything else stays on k_gauss_mfma.  Bit-identical results.
typedef __attribute__((address_space(3))) float lds_f32;
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f32x2v lds_f32x2;
__device__ __forceinline__ unsigned lds_addr(const float* p) { return (unsigned)(size_t)(const lds_f32*)p; }
__device__ __forceinline__ float lds_ld(unsigned a) { return *(const lds_f32*)(size_t)a; }
__device__ __forceinline__ void lds_st(unsigned a, float v) { *(lds_f32*)(size_t)a = v; }
__device__ __forceinline__ void lds_st2(unsigned a, f32x2v v) { *(lds_f32x2*)(size_t)a = v; }

template <int R, int TW>
struct Mfma2Cfg : MfmaCfg<R, TW> {
  static constexpr int D = 2 * R;
  static constexpr int CMIN = -((D + 15) / 16);  // first ring tile a vertical pass reads, relative to H tile it - 1
  static constexpr int NC = -CMIN + 1;
  static constexpr int NT = MfmaCfg<R, TW>::RINGROWS / kMT;  // ring tiles
};

template <int R, int TW>
__global__ __launch_bounds__(kMfmaThreads) void k_gauss_mfma2(ConvArgs a) {
  using C = Mfma2Cfg<R, TW>;
  static_assert((2 * R) % 4 == 0 && C::K == C::KP, "radius must be even");
  static_assert(C::NT >= C::NC + 1, "the ring holds the tiles of one vertical pass beside the tile being written");
  constexpr int TPW = C::TPW, D = C::D, NT = C::NT, CMIN = C::CMIN, NC = C::NC;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float s_mem[];
  const int W = (int)a.w, H = (int)a.h;
  uint32_t sbx, sby;
  strip_block(a.xcdStrips != 0u, sbx, sby);
  const int x0 = (int)sbx * TW;
  const int y0 = (int)sby * (int)a.rowsPerBlock;
  int nrows = (int)a.rowsPerBlock;
  if (y0 + nrows > H) nrows = H - y0;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = (wave ^ (wave >> 2)) & 1, w4 = wave >> 1;  // one wave of each role per SIMD, see k_gauss_mfma
  const int li = lane & 15, lk = lane >> 4;
  float tz[C::KS];
#pragma unroll
  for (int s = 0; s < C::KS; ++s) {
    int t = 4 * s + lk - li;
    int ti = t <= R ? t : 2 * R - t;
    tz[s] = (t >= 0 && t <= 2 * R) ? a.wgt[ti < 0 ? 0 : ti] : 0.0f;
  }
#pragma unroll
  for (int s = 0; s < C::KS; ++s) asm volatile("" : "+v"(tz[s]));
  for (int i = tid; i < (int)(C::ldsBytes / sizeof(float)); i += kMfmaThreads) s_mem[i] = 0.0f;

  // LDS byte addresses, static lane parts (stage [2][kMT][SW] in front, ring [RINGROWS][RSTR] behind)
  constexpr unsigned kStageBytes = (unsigned)kMT * C::SW * 4u, kTileBytes = (unsigned)kMT * C::RSTR * 4u;
  const unsigned ldsBase = lds_addr(s_mem), ringBase = ldsBase + 2u * kStageBytes;
  const unsigned aLane = ldsBase + (unsigned)(li * C::SW + (w4 * TPW) * 16 + (C::RP - R) + lk) * 4u;      // horizontal A reads
  const unsigned hLane = ringBase + (unsigned)((lk * 4) * C::RSTR + (w4 * TPW) * 16 + li) * 4u;          // ring writes
  const unsigned sLane = ldsBase + (unsigned)((4 * w4) * C::SW + C::RP + 4 * lane) * 4u;                  // stage writes, interior
  const unsigned gLane = ldsBase + (unsigned)((4 * w4) * C::SW + (lane < C::RP ? lane : TW + lane)) * 4u;  // stage writes, halo
  const unsigned vLane = ringBase + (unsigned)(lk * C::RSTR + (w4 * TPW) * 16 + li) * 4u;                 // vertical reads

  // ---- horizontal role: staging
  constexpr int NH = 2 * C::RP;
  static_assert(NH <= 64, "one halo float per lane");
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)((uint32_t)W * (uint32_t)H * 4u), 0x00020000);
  const bool haloLane = lane < NH, interiorLane = 4 * lane < TW;
  int hgx = lane < C::RP ? x0 - C::RP + lane : x0 + TW + (lane - C::RP);
  hgx = hgx > W - 1 + R ? W - 1 + R : hgx;
  hgx = hgx < 0 ? -1 - hgx : hgx;
  hgx = hgx > W - 1 ? 2 * W - 1 - hgx : hgx;
  const int vInt = (x0 + 4 * lane) * 4, vHalo = hgx * 4;  // byte offsets inside a row
  u32x4 preI[4];
  unsigned preH[4];
  auto fetch = [&](int s) {
    const int ybase = y0 - R + s * kMT + 4 * w4;  // wave-uniform
    unsigned soff[4];
    if (ybase >= 0 && ybase + 3 < H) {
      const unsigned b = (unsigned)ybase * (unsigned)W * 4u;
#pragma unroll
      for (int k = 0; k < 4; ++k) soff[k] = b + (unsigned)k * (unsigned)W * 4u;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int y = ybase + k;
        y = y > H - 1 + R ? H - 1 + R : y;
        y = y < 0 ? -1 - y : y;
        y = y > H - 1 ? 2 * H - 1 - y : y;
        soff[k] = (unsigned)y * (unsigned)W * 4u;
      }
    }
    if (interiorLane) {
#pragma unroll
      for (int k = 0; k < 4; ++k) preI[k] = __builtin_amdgcn_raw_buffer_load_b128(rin, vInt, (int)soff[k], 0);
    }
    if (haloLane) {
#pragma unroll
      for (int k = 0; k < 4; ++k) preH[k] = __builtin_amdgcn_raw_buffer_load_b32(rin, vHalo, (int)soff[k], 0);
    }
  };
  // sRow: LDS address of this lane's first interior float in stage row 4 w4 of the target buffer; gRow: of its halo float
  auto stage_write = [&](unsigned sRow, unsigned gRow) {
    if (interiorLane) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned e0 = preI[k][0], e1 = preI[k][1], e2 = preI[k][2], e3 = preI[k][3];
        // SW even, RP a multiple of 4: 8-byte aligned pairs
        lds_st2(sRow + (unsigned)(k * C::SW) * 4u, f32x2v{__builtin_bit_cast(float, e0), __builtin_bit_cast(float, e1)});
        lds_st2(sRow + (unsigned)(k * C::SW + 2) * 4u, f32x2v{__builtin_bit_cast(float, e2), __builtin_bit_cast(float, e3)});
      }
    }
    if (haloLane) {
#pragma unroll
      for (int k = 0; k < 4; ++k) lds_st(gRow + (unsigned)(k * C::SW) * 4u, __builtin_bit_cast(float, preH[k]));
    }
  };
  // ---- vertical role: stores
  const unsigned vOut = ((unsigned)li * (unsigned)W + (unsigned)(x0 + (w4 * TPW) * 16 + lk * 4)) * 4u;
  float mn = FLT_MAX, mx = -FLT_MAX;
  constexpr int kNoPending = -(1 << 30);
  f32x4 pend[TPW];
  int pendJ = kNoPending;
  auto store_pending = [&]() {
    const int jbase = pendJ;  // uniform
    char* const orow = reinterpret_cast<char*>(a.out) + (size_t)(long)(y0 + jbase) * (size_t)W * 4u;
    if (a.binOut) {  // uniform; li == lane & 15 has the parity of threadIdx.x
      const int j = jbase + li;
      const bool pairOk = j >= 0 && (j | 1) < nrows;  // this row and its partner (rows 2m, 2m + 1 share a tile: jbase is even)
      const size_t bi = (size_t)((y0 + j) >> 1) * (size_t)(W >> 1) + (size_t)((x0 + (w4 * TPW) * 16 + lk * 4) >> 1);
#pragma unroll
      for (int t4 = 0; t4 < TPW; ++t4) bin2x_from_tile(pend[t4], pairOk, a.binOut, bi + (size_t)t4 * 8);
    }
    if (jbase >= 0 && jbase + kMT <= nrows) {
#pragma unroll
      for (int t4 = 0; t4 < TPW; ++t4) *reinterpret_cast<f32x4*>(orow + vOut + t4 * 64) = pend[t4];
#pragma unroll
      for (int t4 = 0; t4 < TPW; ++t4) {
        const f32x4 v = pend[t4];
        mn = __builtin_fminf(__builtin_fminf(mn, v[0]), v[1]);
        mn = __builtin_fminf(__builtin_fminf(mn, v[2]), v[3]);
        mx = __builtin_fmaxf(__builtin_fmaxf(mx, v[0]), v[1]);
        mx = __builtin_fmaxf(__builtin_fmaxf(mx, v[2]), v[3]);
      }
    } else {
      const int j = jbase + li;
      if (j >= 0 && j < nrows) {
#pragma unroll
        for (int t4 = 0; t4 < TPW; ++t4) {
          const f32x4 v = pend[t4];
          *reinterpret_cast<f32x4*>(orow + vOut + t4 * 64) = v;
          mn = fminf(fminf(mn, v[0]), fminf(v[1], fminf(v[2], v[3])));
          mx = fmaxf(fmaxf(mx, v[0]), fmaxf(v[1], fmaxf(v[2], v[3])));
        }
      }
    }
    pendJ = kNoPending;
  };
  const int steps = (nrows + 2 * R + kMT - 1) / kMT;
  if (role == 0) fetch(0);
  __syncthreads();  // zero fill complete
  if (role == 0) {
    stage_write(sLane, gLane);
    if (steps > 1) fetch(1);
  }
  // Schedule of a step.  Both waves of a SIMD start their MFMA loops right behind the barrier and share the matrix pipe
  // (stamps: the horizontal wave's 80 MFMAs are through after ~3000 cycles, the vertical wave's after ~5500 = 160 slots:
  // the pipe is saturated meanwhile); the horizontal wave's ring / stage / fetch work has no vector instruction and
  // proceeds under the vertical wave's MFMAs.  (s_setprio on either role changed nothing.)
  // Vector instructions of one wave do not issue while the other streams MFMAs (s_memtime: a 20-add address batch of
  // the vertical wave, placed in front of its loop, waited for the END of the horizontal wave's 80 MFMAs, s_setprio or
  // not), so each wave forms the NEXT step's LDS addresses in one batch at the end of its step, behind its last MFMA.
  unsigned aRow = 0, hRow[4] = {0, 0, 0, 0}, sRow = 0, gRow = 0;  // horizontal role: this step's LDS addresses
  unsigned rb[C::KS];  // vertical role: LDS address (lane part included) of the first operand of every k-step of this step
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) rb[ks] = 0;
  auto h_addresses = [&](int it, int tph /* (it - 1) mod NT */) {
    const unsigned par = (unsigned)(it & 1) * kStageBytes;  // uniform
    aRow = aLane + par;
    const unsigned slotH = (unsigned)(tph + 1 == NT ? 0 : tph + 1) * kTileBytes;  // H tile `it` goes to slot it mod NT
#pragma unroll
    for (int r = 0; r < 4; ++r) hRow[r] = hLane + slotH + (unsigned)(r * C::RSTR) * 4u;
    sRow = sLane + (kStageBytes - par);  // the other buffer
    gRow = gLane + (kStageBytes - par);
    asm volatile("" : "+v"(aRow), "+v"(hRow[0]), "+v"(hRow[1]), "+v"(hRow[2]), "+v"(hRow[3]), "+v"(sRow), "+v"(gRow));
  };
  // k-step ks of a vertical pass takes rows (4 ks - D) .. + 3 relative to H tile it - 1, i.e. in-tile row `within` of ring
  // tile c -- both compile-time; the tile's slot is scalar.  Every k-step gets its own address register here, in the batch:
  // with one register per tile and the row as an immediate the compiler pairs the reads into ds_read2_b32, whose 8-bit
  // offsets do not reach the row, and puts a v_add_u32 in front of every k-step -- the vector instruction among MFMAs
  // this kernel exists to avoid.  (Hand-placed ds_read_b32 with 16-bit immediates and explicit s_waitcnt ran at the same
  // speed, 0.233 ms at 65 taps, but a register copy the compiler may insert between such a read and its wait would copy
  // stale data: dropped for the compiler-tracked form.)
  auto v_addresses = [&](int tph /* (it - 1) mod NT */) {
    unsigned tb[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      int slot = tph + CMIN + c;  // uniform; ring tile c (CMIN .. 0, relative to H tile it - 1) sits in slot (tph + c) mod NT
      slot = slot < 0 ? slot + NT : slot;
      tb[c] = (unsigned)slot * kTileBytes;
    }
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      const int rel = 4 * ks - D;  // compile-time: first row of the k-step, relative to H tile it - 1
      const int c = (rel >= 0 ? rel / 16 : -((-rel + 15) / 16));
      const int within = rel - 16 * c;
      rb[ks] = vLane + tb[c - CMIN] + (unsigned)(within * C::RSTR) * 4u;
    }
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) asm volatile("" : "+v"(rb[ks]));
  };
  if (role == 0) {
    h_addresses(0, NT - 1);
  } else {
    v_addresses(0);  // step 1 reads relative to H tile 0
  }
  __syncthreads();
  int tphase = NT - 1;  // ((it - 1) mod NT) for it = 0
  for (int it = 0; it <= steps; ++it) {
    SV_STAMP(0);
    const int tnext = tphase + 1 == NT ? 0 : tphase + 1;  // (it mod NT) = tphase of step it + 1
    if (role == 0) {
      if (it < steps) {
        f32x4 acc[TPW];
#pragma unroll
        for (int t4 = 0; t4 < TPW; ++t4) acc[t4] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        float av[3][TPW];
#pragma unroll
        for (int t4 = 0; t4 < TPW; ++t4) av[0][t4] = lds_ld(aRow + (unsigned)(t4 * 16) * 4u);
        if (C::KS > 1) {
#pragma unroll
          for (int t4 = 0; t4 < TPW; ++t4) av[1][t4] = lds_ld(aRow + (unsigned)(t4 * 16 + 4) * 4u);
        }
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
          if (ks + 2 < C::KS) {
#pragma unroll
            for (int t4 = 0; t4 < TPW; ++t4) av[(ks + 2) % 3][t4] = lds_ld(aRow + (unsigned)(t4 * 16 + 4 * (ks + 2)) * 4u);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t4 = 0; t4 < TPW; ++t4) acc[t4] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks % 3][t4], tz[ks], acc[t4], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        SV_STAMP(1);
#pragma unroll
        for (int t4 = 0; t4 < TPW; ++t4) {
#pragma unroll
          for (int r = 0; r < 4; ++r) lds_st(hRow[r] + (unsigned)(t4 * 16) * 4u, acc[t4][r]);
        }
        SV_STAMP(2);
        if (it + 1 < steps) {
          stage_write(sRow, gRow);
          SV_STAMP(3);
          if (it + 2 < steps) fetch(it + 2);
        }
        SV_STAMP(4);
        if (it + 1 < steps) h_addresses(it + 1, tnext);
      }
    } else {
      const int jbase = (it - 1) * kMT - D;
      if (it >= 1 && jbase + kMT > 0 && jbase < nrows) {  // uniform
#pragma unroll
        for (int t4 = 0; t4 < TPW; ++t4) pend[t4] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        auto loadB = [&](int ks, float (&dst)[TPW]) {
#pragma unroll
          for (int t4 = 0; t4 < TPW; ++t4) dst[t4] = lds_ld(rb[ks] + (unsigned)(t4 * 16) * 4u);
        };
        // B operands are read two k-steps ahead of the MFMAs that use them
        float bv[3][TPW];
        loadB(0, bv[0]);
        if (C::KS > 1) loadB(1, bv[1]);
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
          if (ks + 2 < C::KS) loadB(ks + 2, bv[(ks + 2) % 3]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t4 = 0; t4 < TPW; ++t4) pend[t4] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[ks % 3][t4], tz[ks], pend[t4], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        SV_STAMP(1);
        pendJ = jbase;
        store_pending();
      }
      SV_STAMP(2);
      if (it < steps) v_addresses(tnext);
      SV_STAMP(3);
    }
    tphase = tnext;
    __syncthreads();
  }
  if (a.minmax) block_minmax_commit(mn, mx, a.minmax, s_mem);
}

#include "gauss_rm.inc"
#include "gauss_pair.inc"
#include "gauss_pair_rm.inc"

// ---- S4 for the small levels: one tile per block, no marching ---------------------------------------------------------
// The marching kernels pay a 2R-row warm-up and one barrier per 8 or 16 rows; a 1024^2 or 2048^2 level gives them 64 to
// 512 blocks of 8 to 12 sequential steps each -- 16 to 60 us per level for work the chip does in 2 (octaves 2 and 3: 0.40 ms
// of a 2.2 ms pyramid for 8 % of its pixels).  Here a block owns a 64 x 64 output tile: it stages the (64 + 2R)-row input
// region once (every load of a wave in flight at the same time), its 8 waves run the horizontal pass of the (up to 8) row
// groups side by side and keep the result in registers across a barrier, so that the H rows can go where the input was:
// the tile needs 50 to 67 KB of LDS, two blocks share a CU and one block's loads run under the other's MFMAs.  Then the
// vertical pass of the 16 output tiles, two per wave.  Same banded-Toeplitz MFMA chains in the same k order as
// k_gauss_mfma (the operands come from the same relative LDS positions), so the result is bit-identical.  The halo makes
// the block read 4x and compute 2-3x what it writes: only for levels that are latency-bound anyway (they sit in L2).
template <int R>
struct TileCfg {
  static constexpr int TW = 64, TH = 64;
  static constexpr int K = 16 + 2 * R, KS = (K + 3) / 4, KP = KS * 4, RP = (R + 3) / 4 * 4;
  static constexpr int HR = (TH + 2 * R + 15) / 16 * 16;  // staged input rows = rows of the horizontal pass
  static_assert(HR / 16 <= 8, "one row group per wave");
  static constexpr int CW = TW + 2 * RP + (KP - K);       // staged columns the A operands reach
  static constexpr int SW = CW % 4 == 2 ? CW : CW + 2;    // == 2 (mod 4): 2 x odd (mod 32), conflict-free A reads (see MfmaCfg)
  static_assert(CW % 2 == 0 && SW % 4 == 2, "stage row stride");
  static constexpr int RSTR = TW + 16;                    // == 16 (mod 32)
  static_assert(RSTR <= SW, "the H rows reuse the input's LDS");
  static constexpr int NC = (CW + 63) / 64;               // staged floats per lane and row
  static constexpr int RPW = HR / 8;                      // staged rows per wave
  static constexpr size_t ldsBytes = sizeof(float) * (size_t)HR * SW;
};
constexpr int kTileThreads = 512;

template <int R>
__global__ __launch_bounds__(kTileThreads) void k_gauss_tile(ConvArgs a) {
  using C = TileCfg<R>;
  extern __shared__ __attribute__((aligned(16))) float s_mem[];
  float* s_in = s_mem;  // [HR][SW], then
  float* s_h = s_mem;   // [HR][RSTR] in the same place
  const int W = (int)a.w, H = (int)a.h;
  const int x0 = blockIdx.x * C::TW, y0 = blockIdx.y * C::TH;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  float tz[C::KS];
#pragma unroll
  for (int s = 0; s < C::KS; ++s) {
    int t = 4 * s + lk - li;
    int ti = t <= R ? t : 2 * R - t;
    tz[s] = (t >= 0 && t <= 2 * R) ? a.wgt[ti < 0 ? 0 : ti] : 0.0f;
  }
  // ---- stage rows y0 - R .. y0 - R + HR - 1, columns x0 - RP .. x0 - RP + CW - 1 (mirrored into the image; what lies
  // beyond W - 1 + R / H - 1 + R only meets zero weights or outputs that are not stored and re-reads a valid pixel).
  // Wave w owns rows w * RPW .. + RPW - 1; all of its loads are issued before the first LDS write.
  int gx[C::NC];
#pragma unroll
  for (int i = 0; i < C::NC; ++i) {
    gx[i] = mirror_into(x0 - C::RP + i * 64 + lane, W, R);
  }
  {
    float v[C::RPW][C::NC];
    const int r0 = wave * C::RPW;
#pragma unroll
    for (int k = 0; k < C::RPW; ++k) {
      const int y = mirror_into(y0 - R + r0 + k, H, R);  // wave-uniform
      const float* row = a.in + (size_t)y * W;
#pragma unroll
      for (int i = 0; i < C::NC; ++i) v[k][i] = row[gx[i]];
    }
#pragma unroll
    for (int k = 0; k < C::RPW; ++k) {
#pragma unroll
      for (int i = 0; i < C::NC; ++i)
        if ((i + 1) * 64 <= C::CW || i * 64 + lane < C::CW) s_in[(r0 + k) * C::SW + i * 64 + lane] = v[k][i];
    }
  }
#pragma unroll
  for (int s = 0; s < C::KS; ++s) asm volatile("" : "+v"(tz[s]));
  __syncthreads();
  // ---- horizontal pass: row group `wave` (16 rows) x 4 column tiles, kept in registers until every wave has read its input
  f32x4 acc[4];
#pragma unroll
  for (int t4 = 0; t4 < 4; ++t4) acc[t4] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const bool hasGroup = wave < C::HR / 16;  // uniform
  if (hasGroup) {
    const float* arow = s_in + (16 * wave + li) * C::SW + (C::RP - R) + lk;
    float av[3][4];
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) av[0][t4] = arow[t4 * 16];
    if (C::KS > 1) {
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) av[1][t4] = arow[t4 * 16 + 4];
    }
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      if (ks + 2 < C::KS) {
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) av[(ks + 2) % 3][t4] = arow[t4 * 16 + 4 * (ks + 2)];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) acc[t4] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks % 3][t4], tz[ks], acc[t4], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();  // the input is consumed: its LDS becomes the H rows
  if (hasGroup) {
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      float* dst = s_h + (16 * wave + lk * 4) * C::RSTR + t4 * 16 + li;
      dst[0] = acc[t4][0];
      dst[C::RSTR] = acc[t4][1];
      dst[2 * C::RSTR] = acc[t4][2];
      dst[3 * C::RSTR] = acc[t4][3];
    }
  }
  __syncthreads();
  // ---- vertical pass (transposed, as in k_gauss_mfma): output row tile rt, column tiles 2 (wave & 1) and + 1
  float mn = FLT_MAX, mx = -FLT_MAX;
  {
    const int rt = wave >> 1, ct0 = 2 * (wave & 1);
    int boff[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      int kk = 4 * ks + lk;
      kk = kk > C::K - 1 ? C::K - 1 : kk;
      boff[ks] = (16 * rt + kk) * C::RSTR;
    }
    const float* colp = s_h + ct0 * 16 + li;
    f32x4 pend[2];
    pend[0] = pend[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float bv[3][2];
    bv[0][0] = colp[boff[0]];
    bv[0][1] = colp[boff[0] + 16];
    if (C::KS > 1) {
      bv[1][0] = colp[boff[1]];
      bv[1][1] = colp[boff[1] + 16];
    }
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      if (ks + 2 < C::KS) {
        bv[(ks + 2) % 3][0] = colp[boff[ks + 2]];
        bv[(ks + 2) % 3][1] = colp[boff[ks + 2] + 16];
      }
      __builtin_amdgcn_sched_barrier(0);
      pend[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[ks % 3][0], tz[ks], pend[0], 0, 0, 0);
      pend[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[ks % 3][1], tz[ks], pend[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    const int gy = y0 + 16 * rt + li;
    const bool vec4 = (W & 3) == 0 && (reinterpret_cast<size_t>(a.out) & 15) == 0;
    if (a.binOut) {  // uniform
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const int gxo = x0 + (ct0 + t2) * 16 + lk * 4;
        const bool pairOk = (gy | 1) < H && gxo + 3 < W;  // launch_conv folds the bin only for even W, H with W % 4 == 0
        bin2x_from_tile(pend[t2], pairOk, a.binOut, (size_t)(gy >> 1) * (size_t)(W >> 1) + (size_t)(gxo >> 1));
      }
    }
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const int gxo = x0 + (ct0 + t2) * 16 + lk * 4;
      float* o = a.out + ((size_t)gy * W + gxo);
      const f32x4 v = pend[t2];
      if (gy < H && vec4 && gxo + 3 < W) {
        *reinterpret_cast<f32x4*>(o) = v;
        mn = fminf(fminf(mn, v[0]), fminf(v[1], fminf(v[2], v[3])));
        mx = fmaxf(fmaxf(mx, v[0]), fmaxf(v[1], fmaxf(v[2], v[3])));
      } else if (gy < H) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (gxo + r < W) { o[r] = v[r]; mn = fminf(mn, v[r]); mx = fmaxf(mx, v[r]); }
      }
    }
  }
  if (a.minmax) block_minmax_commit(mn, mx, a.minmax, s_mem);
}

// Generic two-pass fallback for tap counts the pipeline never produces (taps > 65): one 1-D pass per launch.
__global__ __launch_bounds__(256) void k_conv1d(const float* __restrict__ in, float* __restrict__ out, uint32_t w,
                                                uint32_t h, int taps, const float* __restrict__ wgt, int vertical,
                                                float* __restrict__ minmax) {
  uint32_t x = blockIdx.x * 256 + threadIdx.x;
  uint32_t y = blockIdx.y;
  float mn = FLT_MAX, mx = -FLT_MAX;
  if (x < w && y < h) {
    int r = taps / 2;
    float sum = 0.0f;
    for (int k = -r; k <= r; ++k) {
      int sx = vertical ? (int)x : sym_coord((int)x + k, (int)w);
      int sy = vertical ? sym_coord((int)y + k, (int)h) : (int)y;
      sum = __builtin_fmaf(in[(size_t)sy * w + sx], wgt[k + r], sum);
    }
    out[(size_t)y * w + x] = sum;
    mn = sum;
    mx = sum;
  }
  if (minmax) wave_minmax_commit(mn, mx, minmax);
}

// ---- S6: min/max + normalize as stand-alone kernels (kernel-level API) -----------------------------------------------
__global__ __launch_bounds__(256) void k_minmax(const float* __restrict__ in, size_t n, float* __restrict__ minmax) {
  float mn = FLT_MAX, mx = -FLT_MAX;
  size_t stride = (size_t)gridDim.x * 256 * 4;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      float4 v = *reinterpret_cast<const float4*>(in + i);
      mn = fminf(fminf(mn, v.x), fminf(v.y, fminf(v.z, v.w)));
      mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
    } else {
      for (size_t k = i; k < n; ++k) { mn = fminf(mn, in[k]); mx = fmaxf(mx, in[k]); }
    }
  }
  __shared__ float s_red[8];
  block_minmax_commit(mn, mx, minmax, s_red);
}
__global__ __launch_bounds__(256) void k_normalize(float* __restrict__ data, size_t n, const float* __restrict__ minmax) {
  const float mn = minmax[0], mx = minmax[1];
  const sv::Divisor range = sv::make_divisor(mx - mn);
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) data[i] = sv::div_by(data[i] - mn, range);
}

// ---- S6 + S7: both normalisations + DoG, one streaming kernel --------------------------------------------------------
struct DogArgs {
  const float* lvl[svp::kGauss];
  float* dog[svp::kDog];
  const float* lvlMinMax;  // 6 x {min,max}
  float* dogMinMax;        // 5 x {min,max}, nullable
  float* partial;          // PARTIAL mode: float[2 * kDog][kDogMaxWaves], slot = global wave index
  size_t n;
};
// One launch produces the DoG levels first .. last - 1 (wave-uniform kernel arguments): the levels below `first` were
// made by an earlier launch of the split schedule (or emitted by the convolution loaders, see k_gauss_mfma), the ones from
// `last` on follow once their Gaussian levels are complete.  Only the produced levels' {min, max} slots are touched.
template <bool PARTIAL>
__global__ __launch_bounds__(256) void k_dog(DogArgs a, int first, int last) {
  float lmn[svp::kGauss];
  sv::Divisor range[svp::kGauss];  // (v - min) / (max - min) as the IEEE quotient through a shared reciprocal (device_math.h)
#pragma unroll
  for (int b = 0; b < svp::kGauss; ++b) {
    lmn[b] = 0.0f;
    range[b] = sv::Divisor{1.0f, 1.0f};
    if (b >= first && b <= last) {  // levels above `last` may still be in the making
      lmn[b] = a.lvlMinMax[2 * b];
      range[b] = sv::make_divisor(a.lvlMinMax[2 * b + 1] - lmn[b]);
    }
  }
  float dmn[svp::kDog], dmx[svp::kDog];
#pragma unroll
  for (int b = 0; b < svp::kDog; ++b) { dmn[b] = FLT_MAX; dmx[b] = -FLT_MAX; }
  typedef float f32x4nt __attribute__((ext_vector_type(4)));
  auto normalised = [&](const float* lvl, size_t i, float mn, const sv::Divisor& rg) {
    const f32x4nt c0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4nt*>(lvl + i));  // streamed once
    return make_float4(sv::div_by(c0.x - mn, rg), sv::div_by(c0.y - mn, rg), sv::div_by(c0.z - mn, rg), sv::div_by(c0.w - mn, rg));
  };
  size_t stride = (size_t)gridDim.x * 256 * 4;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < a.n; i += stride) {
    float4 prev = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int b = 0; b < svp::kDog; ++b) {
      if (b < first || b >= last) continue;  // uniform
      if (b == first) prev = normalised(a.lvl[b], i, lmn[b], range[b]);
      const float4 cur = normalised(a.lvl[b + 1], i, lmn[b + 1], range[b + 1]);
      float4 d = make_float4(cur.x - prev.x, cur.y - prev.y, cur.z - prev.z, cur.w - prev.w);
      __builtin_nontemporal_store(f32x4nt{d.x, d.y, d.z, d.w}, reinterpret_cast<f32x4nt*>(a.dog[b] + i));
      dmn[b] = fminf(fminf(dmn[b], d.x), fminf(d.y, fminf(d.z, d.w)));
      dmx[b] = fmaxf(fmaxf(dmx[b], d.x), fmaxf(d.y, fmaxf(d.z, d.w)));
      prev = cur;
    }
  }
  if (PARTIAL) {
    // no LDS, no atomics: every wave leaves its {min, max} in its own slot and k_dog_finalize reduces them.  Blocks of
    // this mode are short-lived (large grids), so that convolution blocks queued on another stream find room on the CUs.
    const int lane = threadIdx.x & 63;
    const size_t slot = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
#pragma unroll
    for (int b = 0; b < svp::kDog; ++b) {
      if (b < first || b >= last) continue;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        dmn[b] = fminf(dmn[b], __shfl_xor(dmn[b], o, 64));
        dmx[b] = fmaxf(dmx[b], __shfl_xor(dmx[b], o, 64));
      }
      if (lane == 0) {
        a.partial[(size_t)(2 * b) * svp::kDogMaxWaves + slot] = dmn[b];
        a.partial[(size_t)(2 * b + 1) * svp::kDogMaxWaves + slot] = dmx[b];
      }
    }
    return;
  }
  if (a.dogMinMax) {
    // the {min, max} pairs in one block reduction: wave shuffles, one barrier, up to ten lanes finish and commit
    __shared__ float s_red[4][2 * svp::kDog];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int b = 0; b < svp::kDog; ++b) {
      if (b < first || b >= last) continue;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        dmn[b] = fminf(dmn[b], __shfl_xor(dmn[b], o, 64));
        dmx[b] = fmaxf(dmx[b], __shfl_xor(dmx[b], o, 64));
      }
      if (lane == 0) { s_red[wave][2 * b] = dmn[b]; s_red[wave][2 * b + 1] = dmx[b]; }
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 2 * svp::kDog && (t >> 1) >= first && (t >> 1) < last) {
      float v = s_red[0][t];
      for (int w = 1; w < 4; ++w) v = (t & 1) ? fmaxf(v, s_red[w][t]) : fminf(v, s_red[w][t]);
      if (t & 1) atomic_max_f(a.dogMinMax + t, v);
      else atomic_min_f(a.dogMinMax + t, v);
    }
  }
}

// PARTIAL mode, second step: block b reduces the per-wave partials of DoG level first + b and stores its {min, max}.
// 1024 threads: the kernel sits between the pass of octave o and that of octave o + 1 on the side stream, and five blocks
// of 256 threads walking 14 336 partials each took 44 us of pure load latency there.
constexpr int kFinalizeThreads = 1024;
__global__ __launch_bounds__(kFinalizeThreads) void k_dog_finalize(const float* __restrict__ partial, unsigned waves, int first,
                                                                   float* __restrict__ dogMinMax) {
  const int b = first + (int)blockIdx.x;
  const float* pmn = partial + (size_t)(2 * b) * svp::kDogMaxWaves;
  const float* pmx = partial + (size_t)(2 * b + 1) * svp::kDogMaxWaves;
  float mn = FLT_MAX, mx = -FLT_MAX;
  for (unsigned i = threadIdx.x; i < waves; i += kFinalizeThreads) {
    mn = fminf(mn, pmn[i]);
    mx = fmaxf(mx, pmx[i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o, 64));
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  }
  __shared__ float s_red[2 * kFinalizeThreads / 64];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_red[2 * wave] = mn; s_red[2 * wave + 1] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kFinalizeThreads / 64; ++w) { mn = fminf(mn, s_red[2 * w]); mx = fmaxf(mx, s_red[2 * w + 1]); }
    dogMinMax[2 * b] = mn;
    dogMinMax[2 * b + 1] = mx;
  }
}

// ---- S6 + S7 + S8 without the DoG levels: the fused pass of ssrlcv_hip_sift_build_dog --------------------------------------
// The reference writes five DoG levels per octave (subtractImages), reads them back to find their min / max, then reads
// them three more times (extrema, gradients, refinement).  The pipeline keeps the six Gaussian levels instead and never
// writes a DoG level: this pass streams the Gaussian levels once, forms DoG[b] = N(level b+1) - N(level b) in registers
// (the arithmetic of k_dog, so the same values), reduces the DoG levels' {min, max} (the second normalisation of
// findKeyPoints needs them, src/FeatureFactory.cu:472) and runs findExtrema (src/FeatureFactory.cu:847-882) on the values
// while they are there: a pixel of DoG level b = 1..3 is flagged when it equals the max or the min of its 3x3x3
// neighbourhood (non-strict).  Every later consumer (refinement, edge test, gradient tables) evaluates the DoG values it
// samples from the two Gaussian levels again.  Per octave pixel: 24 B read + 1 B written, against 24 + 20 of k_dog and
// another 20 + 1 of a separate extrema pass.
//
// A wave owns a strip of 62 * NPX columns and marches down `rowsPerWave` rows of it (plus one halo row above and below);
// a lane holds NPX consecutive pixels, lanes 0 and 63 the strip's halo columns (strips overlap by two lanes: 3 % more
// loads and arithmetic, against ~15 % for evaluating the halo columns in two masked lanes -- the pass is VALU bound -- and
// the strip starts stay 4 NPX-byte aligned).  The 27-value max / min is separable and evaluated in the order columns (the
// neighbours' pixels come by DPP wave shifts) -> levels -> rows, so the state carried from row to row is the
// level-combined 3-wide max / min of the two previous rows (3 levels x 2 x NPX registers each) and the centre values of
// the previous row.  max and min are exact: the grouping changes no flag.
// flags[p]: bit k = extremum of DoG level k + 1; bit kNoiseFlagShift + k = the same and |value| >= minAbs (the first
// removeNoise, which tests the raw DoG value).
// One launch covers the DoG levels first .. last - 1: flags for the levels first + 1 .. last - 2, {min, max} for
// mmFirst .. last - 1.  orFlags: the flag byte already holds the bits of an earlier launch of the split schedule.
// (These four are template parameters: as wave-uniform kernel arguments they cut the row loop into fifty basic blocks
// that the scheduler could not move loads across.)
struct DogxArgs {
  const float* lvl[svp::kGauss];
  const float* lvlMinMax;  // 6 x {min, max}
  uint8_t* flags;
  float* partial;          // float[2 * kDog][kDogMaxWaves]: per-wave {min, max} partials, reduced by k_dog_finalize
  int w, h;
  int strips;              // ceil(w / (62 * NPX))
  int rowsPerWave;
  float minAbs;
};

__device__ __forceinline__ float dpp_from_lane_below(float v) {  // lane i <- lane i - 1 (wave_shr:1); lane 0 keeps its own
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_from_lane_above(float v) {  // lane i <- lane i + 1 (wave_shl:1); lane 63 keeps its own
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, false));
}
__device__ __forceinline__ float dmax3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float dmin3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }

template <int NPX, int first, int last, int mmFirst, bool orFlags>
#ifndef SSRLCV_DOGX_MINWAVES
#define SSRLCV_DOGX_MINWAVES 1
#endif
#ifndef SSRLCV_DOGX_VGPR_MIN
#define SSRLCV_DOGX_VGPR_MIN 0
#endif
#ifndef SSRLCV_EARLY_CHAIN0
#define SSRLCV_EARLY_CHAIN0 1  // mask of the octaves whose list chain a fused extract starts from inside build_dog
#endif
#ifndef SSRLCV_DOGX_RING
#define SSRLCV_DOGX_RING 1
#endif
#ifndef SSRLCV_PAIR_F32_MINPX
#define SSRLCV_PAIR_F32_MINPX (~(size_t)0)  // float-sourced octaves keep their two launches (see build_dog)
#endif
__global__ __launch_bounds__(256, SSRLCV_DOGX_MINWAVES) void k_dogx(DogxArgs a) {
  typedef float vec __attribute__((ext_vector_type(NPX)));
  const int lane = threadIdx.x & 63;
  const int gwave = blockIdx.x * 4 + (threadIdx.x >> 6);  // adjacent waves = adjacent strips of the same rows
  const int strip = gwave % a.strips, seg = gwave / a.strips;
  const int W = a.w, H = a.h;
  const int r0 = seg * a.rowsPerWave;
  if (r0 >= H) return;  // wave-uniform
  const int r1 = r0 + a.rowsPerWave < H ? r0 + a.rowsPerWave : H;
  const int x = (strip * 62 + lane - 1) * NPX;     // first column of this lane (lane 0 of strip 0: -NPX)
  const bool mine = lane >= 1 && lane <= 62 && x < W;  // W is a multiple of NPX: a lane is inside or outside as a whole
  // lanes outside the image re-read the nearest pixels: they only ever neighbour border pixels, which do not flag, and
  // what they add to the {min, max} are values of real pixels
  const int xl = x < 0 ? 0 : (x < W ? x : W - NPX);
  float lmn[svp::kGauss];
  sv::Divisor rg[svp::kGauss];
#pragma unroll
  for (int b = 0; b < svp::kGauss; ++b) {
    lmn[b] = 0.0f;
    rg[b] = sv::Divisor{1.0f, 1.0f};
    if (b >= first && b <= last) {
      lmn[b] = a.lvlMinMax[2 * b];
      rg[b] = sv::make_divisor(a.lvlMinMax[2 * b + 1] - lmn[b]);
#if SSRLCV_DOGX_VGPR_MIN
      asm volatile("" : "+v"(lmn[b]));  // a vector source: a subtraction with a scalar source issues a cycle later (tools/valu_rate.hip)
#endif
    }
  }
  float dmn[svp::kDog], dmx[svp::kDog];
#pragma unroll
  for (int b = 0; b < svp::kDog; ++b) { dmn[b] = FLT_MAX; dmx[b] = -FLT_MAX; }
  // carried state: level-combined 3-wide max / min of rows y - 2 and y - 1, centre values of row y - 1 (levels 1..3).
  // Three register sets take the roles {row y - 2, row y - 1, row y} in rotation and the row loop is unrolled three
  // times, so that a row's result is formed in the set whose row has just gone out of use: carried as "A = B, B = new"
  // the state cost 60 v_mov_b32 of the row's 430 vector instructions (the pass is VALU bound).
  float gmxA[3][NPX], gmnA[3][NPX], gmxB[3][NPX], gmnB[3][NPX], gmxC[3][NPX], gmnC[3][NPX];
  float ctrX[3][NPX], ctrY[3][NPX], ctrZ[3][NPX];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      gmxA[k][i] = gmnA[k][i] = gmxB[k][i] = gmnB[k][i] = gmxC[k][i] = gmnC[k][i] = 0.0f;
      ctrX[k][i] = ctrY[k][i] = ctrZ[k][i] = 0.0f;
    }
  // The next row is brought into the registers of the current one as soon as its values have been normalised (they are
  // dead then).  (A separate `next` set, copied at the top of the loop, was 24 of the row's 430 vector instructions.)
  // NPX < 4 (rows that are not 16-byte aligned): by plain loads, one row ahead.
  // NPX == 4: one row ahead was not enough -- a wave spends ~0.7 us on a row's arithmetic, two waves fit a SIMD (241
  // registers) and a loaded row arrives after ~2 us: the pass ran at 2.7 us per row and wave, and taking 60 instructions
  // out of the row's 430 changed nothing (round 5).  The rows y + 2 and y + 3 are therefore in flight as LDS-DMA transfers
  // (global_load_lds: no registers) into a two-slot ring per wave, and row y + 1 is read from its slot where the plain
  // load used to be issued; its slot then takes row y + 3.  6 KB per slot, 48 KB per block, two blocks per CU.
  constexpr bool RING = SSRLCV_DOGX_RING && NPX == 4;
  constexpr int NL = last - first + 1;
  static_assert(!RING || (NL >= 4 && NL <= 6), "ring_take's wait counts are written out for 4..6 levels");
  __shared__ float4 s_ring[RING ? 4 * 2 * NL * 64 : 1];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float4* const ring = s_ring + (RING ? wave * 2 * NL * 64 : 0);
  vec cur[svp::kGauss];
  auto fetch = [&](int y) {
    y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);  // rows -1 and H only neighbour border rows, which do not flag
    const size_t row = (size_t)y * W + xl;
#pragma unroll
    for (int b = first; b <= last; ++b) cur[b] = __builtin_nontemporal_load(reinterpret_cast<const vec*>(a.lvl[b] + row));
  };
  auto ring_issue = [&](int y, int slot) {  // row y (clamped) -> slot; lane l's 16 bytes land at slot base + 16 l
    y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
    const size_t row = (size_t)y * W + xl;
#pragma unroll
    for (int b = first; b <= last; ++b)
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(a.lvl[b] + row),
                                       (void __attribute__((address_space(3)))*)(ring + (slot * NL + (b - first)) * 64), 16, 0, 2 /* nt */);
  };
  // The slot is read by ds_read_b128 in inline assembly: a read the compiler sees makes it wait for EVERY transfer in
  // flight, not only this slot's.  vmcnt(NL): all but the NL youngest vector-memory operations have completed -- the NL
  // transfers of the following row were issued behind this row's, and loads complete in order (flag stores in between only
  // make the wait stricter); vmcnt(0) where no row follows.
  auto ring_take = [&](int slot, bool followed) {
    if (followed) {  // (NL transfers per row)
      if (NL == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (NL == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const uint32_t addr = (uint32_t)(uintptr_t)(void __attribute__((address_space(3)))*)(ring + slot * NL * 64) + 16u * (uint32_t)lane;
#pragma unroll
    for (int b = first; b <= last; ++b)
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(cur[b]) : "v"(addr), "n"((b - first) * 1024) : "memory");
    // (the values are outputs of the wait: no use of them can be scheduled in front of it)
    if (NL == 6) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[first]), "+v"(cur[first + 1]), "+v"(cur[first + 2]), "+v"(cur[first + 3]), "+v"(cur[first + 4]), "+v"(cur[last]) : : "memory");
    else if (NL == 5) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[first]), "+v"(cur[first + 1]), "+v"(cur[first + 2]), "+v"(cur[first + 3]), "+v"(cur[last]) : : "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[first]), "+v"(cur[first + 1]), "+v"(cur[first + 2]), "+v"(cur[last]) : : "memory");
  };
  const int yEnd = r0 - 1 + (r1 - r0 + 2 + 2) / 3 * 3 - 1;  // last row of the last trip of three (>= r1)
  unsigned pendPacked = 0;
  unsigned* pendPtr = nullptr;  // (per lane; null: nothing pending)
  auto flush_flags = [&]() {
    if (pendPtr) *pendPtr = orFlags ? (*pendPtr | pendPacked) : pendPacked;
    pendPtr = nullptr;
  };
  // one row: (oMx, oMn) = state of row y - 2, (pMx, pMn) of row y - 1, (nMx, nMn) receives row y's; cPrev = centre values
  // of row y - 1, cNew receives row y's
  auto row = [&](int y, const float (&oMx)[3][NPX], const float (&oMn)[3][NPX], const float (&pMx)[3][NPX],
                 const float (&pMn)[3][NPX], float (&nMx)[3][NPX], float (&nMn)[3][NPX], const float (&cPrev)[3][NPX],
                 float (&cNew)[3][NPX]) __attribute__((always_inline)) {
    // normalised levels and DoG values of this row
    float d[svp::kDog][NPX];
    {
      float prev[NPX];
#pragma unroll
      for (int b = first; b <= last; ++b) {
        float n[NPX];
#pragma unroll
        for (int i = 0; i < NPX; ++i) n[i] = sv::div_by(cur[b][i] - lmn[b], rg[b]);
        if (b > first) {
#pragma unroll
          for (int i = 0; i < NPX; ++i) d[b - 1][i] = n[i] - prev[i];
        }
#pragma unroll
        for (int i = 0; i < NPX; ++i) prev[i] = n[i];
      }
    }
    // (the compiler would issue these loads at the top of the row, into other registers, and copy: the DoG values are
    // pinned in front of them)
#pragma unroll
    for (int b = first; b < last; ++b) {
#pragma unroll
      for (int i = 0; i < NPX; ++i) asm volatile("" : "+v"(d[b][i]) : : "memory");
    }
    if (RING) {
      if (y < yEnd) {  // (wave-uniform)
        const int slot = (y + 1 - (r0 - 1)) & 1;
        ring_take(slot, y + 2 <= yEnd);
        flush_flags();
        if (y + 3 <= yEnd) ring_issue(y + 3, slot);
      } else {
        flush_flags();
      }
    } else {
      fetch(y + 1);  // (clamped: the row behind the last one is loaded and not used)
    }
    // 3-wide max / min per level; the level's {min, max} ride on them (pixel i's triple covers i - 1 .. i + 1; the halo
    // rows and columns a wave sees beyond its own are pixels of the image too)
    float hmx[svp::kDog][NPX], hmn[svp::kDog][NPX];
#pragma unroll
    for (int b = first; b < last; ++b) {
      const float left = dpp_from_lane_below(d[b][NPX - 1]);
      const float right = dpp_from_lane_above(d[b][0]);
#pragma unroll
      for (int i = 0; i < NPX; ++i) {
        const float l = i == 0 ? left : d[b][i - 1], r = i == NPX - 1 ? right : d[b][i + 1];
        hmx[b][i] = dmax3(l, d[b][i], r);
        hmn[b][i] = dmin3(l, d[b][i], r);
      }
      if (b >= mmFirst) {
        if (NPX == 4) {
          dmx[b] = dmax3(dmx[b], hmx[b][1], d[b][3]);
          dmn[b] = dmin3(dmn[b], hmn[b][1], d[b][3]);
        } else if (NPX == 2) {
          dmx[b] = dmax3(dmx[b], d[b][0], d[b][1]);
          dmn[b] = dmin3(dmn[b], d[b][0], d[b][1]);
        } else {
          dmx[b] = __builtin_fmaxf(dmx[b], d[b][0]);
          dmn[b] = __builtin_fminf(dmn[b], d[b][0]);
        }
      }
    }
    // ... then across the levels k, k + 1, k + 2 (DoG level k + 1 and its two neighbours)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
      for (int i = 0; i < NPX; ++i) nMx[k][i] = nMn[k][i] = 0.0f;
      if (k < first || k + 2 >= last) continue;
#pragma unroll
      for (int i = 0; i < NPX; ++i) {
        nMx[k][i] = dmax3(hmx[k][i], hmx[k + 1][i], hmx[k + 2][i]);
        nMn[k][i] = dmin3(hmn[k][i], hmn[k + 1][i], hmn[k + 2][i]);
      }
    }
    // flags of row y - 1 (its three rows y - 2, y - 1, y are complete)
    const int yo = y - 1;
    if (yo >= r0 && yo < r1 && mine) {  // (uniform but for `mine`)
      const bool rowIn = yo > 0 && yo < H - 1;
      unsigned packed = 0;
#pragma unroll
      for (int i = 0; i < NPX; ++i) {
        const bool in = rowIn && (x + i) > 0 && (x + i) < W - 1;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          if (k < first || k + 2 >= last) continue;
          const float hi = dmax3(oMx[k][i], pMx[k][i], nMx[k][i]);
          const float lo = dmin3(oMn[k][i], pMn[k][i], nMn[k][i]);
          const float c = cPrev[k][i];
          const bool ext = in && (hi == c || lo == c);
          const bool loud = ext && !(__builtin_fabsf(c) < a.minAbs);
          packed |= (ext ? 1u << (8 * i + k) : 0u) | (loud ? 1u << (8 * i + svp::kNoiseFlagShift + k) : 0u);
        }
      }
      uint8_t* fp = a.flags + (size_t)yo * W + x;
      if (RING) {
        // stored by the NEXT row, in front of its transfers: a store between two rows' transfers would be among the six
        // youngest operations ring_take leaves in flight and make it wait for part of the row behind the one it needs
        pendPacked = packed;
        pendPtr = reinterpret_cast<unsigned*>(fp);
      } else if (NPX == 4) {
        unsigned* p4 = reinterpret_cast<unsigned*>(fp);
        *p4 = orFlags ? (*p4 | packed) : packed;
      } else if (NPX == 2) {
        unsigned short* p2 = reinterpret_cast<unsigned short*>(fp);
        *p2 = (unsigned short)(orFlags ? (*p2 | packed) : packed);
      } else {
        *fp = (uint8_t)(orFlags ? (*fp | packed) : packed);
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
      for (int i = 0; i < NPX; ++i) cNew[k][i] = (k < first || k + 2 >= last) ? 0.0f : d[k + 1][i];
    }
  };
  // rows r0 - 1 .. r1, rounded up to whole trips of three: the rows behind r1 write no flag, and what they add to the
  // {min, max} are values of real pixels (launch_dogx makes rowsPerWave + 2 a multiple of 3: only a last segment pays)
  if (RING) {
    ring_issue(r0 - 1, 0);
    ring_issue(r0, 1);
    ring_take(0, true);
    ring_issue(r0 + 1, 0);
  } else {
    fetch(r0 - 1);
  }
  for (int y = r0 - 1; y <= r1; y += 3) {
    row(y, gmxA, gmnA, gmxB, gmnB, gmxC, gmnC, ctrX, ctrY);
    row(y + 1, gmxB, gmnB, gmxC, gmnC, gmxA, gmnA, ctrY, ctrZ);
    row(y + 2, gmxC, gmnC, gmxA, gmnA, gmxB, gmnB, ctrZ, ctrX);
  }
  if (RING) flush_flags();
  if (RING) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (every transfer issued has been taken; a wave must not end with one in flight)
  // per-wave {min, max} partials (k_dog_finalize reduces them: no atomics)
#pragma unroll
  for (int b = mmFirst; b < last; ++b) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      dmn[b] = fminf(dmn[b], __shfl_xor(dmn[b], o, 64));
      dmx[b] = fmaxf(dmx[b], __shfl_xor(dmx[b], o, 64));
    }
    if (lane == 0) {
      a.partial[(size_t)(2 * b) * svp::kDogMaxWaves + gwave] = dmn[b];
      a.partial[(size_t)(2 * b + 1) * svp::kDogMaxWaves + gwave] = dmx[b];
    }
  }
}

// launches k_dogx + k_dog_finalize for the DoG levels first .. last - 1 of one octave (see k_dogx)
int launch_dogx(const float* const levels[svp::kGauss], const float* levelMinMax, uint32_t w, uint32_t h, uint8_t* flags,
                float* dogMinMax, float* partial, int first, int last, int mmFirst, bool orFlags, float minAbs,
                unsigned targetWaves, hipStream_t st) {
  if (first < 0 || last > svp::kDog || last - first < 1 || mmFirst < first || mmFirst >= last) return SSRLCV_ERR_INVALID_ARG;
  DogxArgs a;
  for (int b = 0; b < svp::kGauss; ++b) a.lvl[b] = levels[b];
  a.lvlMinMax = levelMinMax;
  a.flags = flags;
  a.partial = partial;
  a.w = (int)w;
  a.h = (int)h;
  a.minAbs = minAbs;
  static const int forceNpx = svdev::env("SSRLCV_DOGX_NPX") ? atoi(svdev::env("SSRLCV_DOGX_NPX")) : 0;
  bool aligned16 = (w & 3) == 0, aligned8 = (w & 1) == 0;
  for (int b = first; b <= last; ++b) {
    aligned16 = aligned16 && (reinterpret_cast<size_t>(levels[b]) & 15) == 0;
    aligned8 = aligned8 && (reinterpret_cast<size_t>(levels[b]) & 7) == 0;
  }
  aligned16 = aligned16 && (reinterpret_cast<size_t>(flags) & 3) == 0;
  aligned8 = aligned8 && (reinterpret_cast<size_t>(flags) & 1) == 0;
  // pixels per lane: 4 where the rows are 16-byte aligned (2.15 vector instructions per pixel against 2.7 with 2; 1.89 against
  // 1.96 ms per 4096^2 build_dog); SSRLCV_DOGX_NPX=1 / 2 force fewer
  int npx = aligned16 ? 4 : aligned8 ? 2 : 1;
  if (forceNpx == 2 && aligned8) npx = 2;
  if (forceNpx == 1) npx = 1;
  a.strips = (int)((w + 62 * npx - 1) / (62 * npx));
  if (targetWaves > svp::kDogMaxWaves) targetWaves = svp::kDogMaxWaves;
  // rows per wave: about targetWaves waves in the launch, at least 16 rows (2 halo rows per segment)
  unsigned segs = targetWaves / (unsigned)a.strips;
  if (segs < 1) segs = 1;
  unsigned rows = (h + segs - 1) / segs;
  if (rows < 16) rows = 16;
  rows += (3 - (rows + 2) % 3) % 3;  // the kernel's row loop runs in trips of three rows over rows + 2
  a.rowsPerWave = (int)rows;
  segs = (h + rows - 1) / rows;
  const unsigned waves = segs * (unsigned)a.strips;
  if (waves > svp::kDogMaxWaves) return SSRLCV_ERR_UNSUPPORTED;
  const dim3 grid((waves + 3) / 4);
  // waves of the last block beyond `waves` return at once (r0 >= H) without touching their partial slot: finalize reads
  // `waves` slots only
#define SSRLCV_LAUNCH_DOGX(F, L, M, O)                                                               \
  do {                                                                                              \
    if (npx == 4) hipLaunchKernelGGL((k_dogx<4, F, L, M, O>), grid, dim3(256), 0, st, a);           \
    else if (npx == 2) hipLaunchKernelGGL((k_dogx<2, F, L, M, O>), grid, dim3(256), 0, st, a);      \
    else hipLaunchKernelGGL((k_dogx<1, F, L, M, O>), grid, dim3(256), 0, st, a);                    \
  } while (0)
  if (first == 0 && last == 5 && mmFirst == 0 && !orFlags) SSRLCV_LAUNCH_DOGX(0, 5, 0, false);
  else if (first == 0 && last == 3 && mmFirst == 0 && !orFlags) SSRLCV_LAUNCH_DOGX(0, 3, 0, false);
  else if (first == 1 && last == 5 && mmFirst == 3 && orFlags) SSRLCV_LAUNCH_DOGX(1, 5, 3, true);
  else return SSRLCV_ERR_INVALID_ARG;  // the three launches build_dog's schedule is made of
#undef SSRLCV_LAUNCH_DOGX
  hipLaunchKernelGGL(k_dog_finalize, dim3((unsigned)(last - mmFirst)), dim3(kFinalizeThreads), 0, st, partial, waves, mmFirst, dogMinMax);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

// u8src (nullable): the convolution's input is the 2x upsample of this u8 image (w/2 x h/2) and `in` is not read; only
// honoured where upsample_fusable() says so (row-staged VALU kernel on full, aligned strips)
bool upsample_fusable(uint32_t w, uint32_t h, int taps) {
  static const bool off = svdev::env("SSRLCV_NO_UPSAMPLE_FUSION") != nullptr || svdev::env("SSRLCV_GAUSS_MFMA") != nullptr;
  return !off && taps / 2 <= 8 && w % kTX == 0 && (w & 3) == 0 && (h & 1) == 0;
}

// VALU formulation: full 256-column strips with 16-byte aligned rows take the row-staged kernel, a partial last strip (or
// everything, when the rows are not aligned) the generic one.  firstColumn > 0: only the strips from that column on (the
// remainder of a launch whose full strips went to the MFMA kernel).
void launch_valu(ConvArgs a, int RT, uint32_t firstColumn, hipStream_t st) {
  const uint32_t w = a.w, h = a.h;
  // rows per block: aim for >= 1024 blocks, never below 64 rows (halo recompute = 2R / rows).  Sizing the launch to
  // whole rounds of resident blocks (768 slots) was tried and measured no different on MI355X.
  const uint32_t bx = (w - firstColumn + kTX - 1) / kTX;
  uint32_t rows = h;
  while (rows > 64 && ((w + kTX - 1) / kTX) * ((h + rows - 1) / rows) < 1024) rows = (rows + 1) / 2;
  rows = (rows + kNR - 1) / kNR * kNR;
  a.rowsPerBlock = rows;
  const uint32_t by = (h + rows - 1) / rows;
  const bool aligned = firstColumn == 0 && (w & 3) == 0 && (a.u8 || (reinterpret_cast<size_t>(a.in) & 15) == 0);
  const uint32_t nFull = aligned ? w / kTX : 0;
#define SSRLCV_LAUNCH_VALU(RR)                                                                                 \
  do {                                                                                                          \
    if (nFull && a.u8) hipLaunchKernelGGL((k_gauss_strip<RR, true>), dim3(nFull, by), dim3(kTX), 0, st, a);     \
    else if (nFull) hipLaunchKernelGGL((k_gauss_strip<RR, false>), dim3(nFull, by), dim3(kTX), 0, st, a);       \
    if (nFull < bx) {                                                                                           \
      a.x0base = firstColumn + nFull * kTX;                                                                     \
      hipLaunchKernelGGL(k_gauss_fused<RR>, dim3(bx - nFull, by), dim3(kTX), 0, st, a);                         \
    }                                                                                                           \
  } while (0)
  switch (RT) {
    case 6: SSRLCV_LAUNCH_VALU(6); break;
    case 8: SSRLCV_LAUNCH_VALU(8); break;
    case 11: SSRLCV_LAUNCH_VALU(11); break;
    case 16: SSRLCV_LAUNCH_VALU(16); break;
    case 23: SSRLCV_LAUNCH_VALU(23); break;
    default: SSRLCV_LAUNCH_VALU(32); break;
  }
#undef SSRLCV_LAUNCH_VALU
}

// Which radii go to the register-marching kernel (gauss_rm.inc): a bit per padded radius 6, 8, 12, 16, 24, 32, and the
// smallest level (pixels) it is used for.  Measured per 8192^2 level on MI355X (tools/gauss_rm_lab.hip, us, established
// kernel -> register-marching): 13 taps 114 -> 123, 17 taps 117 -> 123, 23 taps 154 -> 126, 33 taps 169 -> 154, 47 taps
// 207 -> 186, 65 taps 240 -> 268; per 4096^2 level 41 -> 42, 45 -> 53, 56 -> 62 at 23 / 33 / 47 taps (its blocks are 128
// rows there, a quarter to a half of them warm-up) -- but inside build_dog, where octave 1 runs beside the DoG / extrema pass
// of octave 0, the stage is 10-20 us shorter with them than without (median of 30, three alternations: 1.766-1.783 against
// 1.787-1.797 ms).  Default: radii 12, 16 and 24 from 2^24 pixels up.
// SSRLCV_GAUSS_RM=<mask>, SSRLCV_GAUSS_RM_MINPX=<pixels>, SSRLCV_GAUSS_RM_ROWS=<rows per block> override; lab drivers
// overwrite them in place.  (The first level of octave 0 -- u8 upsample in the loader -- was built on this kernel too:
// bit-identical, and the stage took the same 1.76-1.77 ms as with k_gauss_strip<6, true>; not kept.)
int& rm_mask() {
  static int m = svdev::env("SSRLCV_GAUSS_RM") ? atoi(svdev::env("SSRLCV_GAUSS_RM")) : (4 | 8 | 16);
  return m;
}
size_t& rm_min_px() {
  static size_t v = svdev::env("SSRLCV_GAUSS_RM_MINPX") ? (size_t)atoll(svdev::env("SSRLCV_GAUSS_RM_MINPX")) : ((size_t)1 << 24);
  return v;
}
// which radii take the one-barrier form of the register-marching kernel (two H buffers, see RmCfg): same bits as rm_mask()
int& rm_oneb_mask() {
  static int m = svdev::env("SSRLCV_GAUSS_RM_ONEB") ? atoi(svdev::env("SSRLCV_GAUSS_RM_ONEB")) : 0;
  return m;
}
int& rm_rows() {
  static int r = svdev::env("SSRLCV_GAUSS_RM_ROWS") ? atoi(svdev::env("SSRLCV_GAUSS_RM_ROWS")) : 0;
  return r;
}

// binOut (nullable): where the 2x2 bin of the result may be written by the convolution itself; *binned tells whether it was
// (only k_gauss_mfma2 / k_gauss_tile do it, for even sizes with W % 4 == 0 and no partial strip) -- otherwise the caller
// runs k_bin2x
int launch_conv(const float* in, float* out, float* tmp, uint32_t w, uint32_t h, int taps, const float* weights_host,
                float* minmax, hipStream_t st, const uint8_t* u8src = nullptr, float* binOut = nullptr,
                bool* binned = nullptr) {
  if (taps < 1 || (taps & 1) == 0 || taps > svp::kMaxTaps) return SSRLCV_ERR_INVALID_ARG;
  int R = taps / 2;
  if (R > 32) return SSRLCV_ERR_UNSUPPORTED;  // the pipeline's sigma ladder never exceeds 65 taps
  ConvArgs a;
  a.in = in;
  a.out = out;
  a.minmax = minmax;
  a.w = w;
  a.h = h;
  a.x0base = 0;
  a.rowsPerBlock = h;
  a.binOut = nullptr;
  static const bool noXcdStrips = svdev::env("SSRLCV_NO_XCD_STRIPS") != nullptr;  // (developer build: the plain block order)
  a.xcdStrips = noXcdStrips ? 0u : 1u;
  if (binned) *binned = false;
  static const bool noBinFold = svdev::env("SSRLCV_NO_BIN_FUSION") != nullptr;
  const bool canBin = binOut && !noBinFold && (w & 3) == 0 && (h & 1) == 0 && (reinterpret_cast<size_t>(binOut) & 7) == 0;
#ifdef SSRLCV_STAMPS
  a.stamps = g_lab_stamps;
#endif
  a.u8 = u8src;
  if (u8src && !upsample_fusable(w, h, taps)) return SSRLCV_ERR_INVALID_ARG;
  memset(a.wgt, 0, sizeof a.wgt);
  // pad the tap set symmetrically into the smallest templated radius: extra taps carry weight 0 and do not change the
  // fmaf chain (0*x + s is exact, so the result is identical)
  int RT = R <= 6 ? 6 : R <= 8 ? 8 : R <= 11 ? 11 : R <= 16 ? 16 : R <= 23 ? 23 : 32;
  for (int k = 0; k <= R; ++k) {
    if (weights_host[k] != weights_host[taps - 1 - k]) return SSRLCV_ERR_UNSUPPORTED;  // symmetric taps only
    a.wgt[(RT - R) + k] = weights_host[k];
  }
  // Two bit-identical formulations.  Measured per 16384^2 level on MI355X (ms, taps 13/17/23/33/47/65): VALU marching
  // kernel (row-staged strips) 0.464/0.517/0.690/0.654/1.120/1.102, f32-MFMA banded Toeplitz -/-/0.613/0.669/0.770/0.875
  // -- the band wastes (16 + 2R - taps) / (16 + 2R) of the matrix pipe, more than half at R = 6, and the VALU kernel is
  // within 12 % of copy speed there.  Default: MFMA from 23 taps up.  SSRLCV_GAUSS_VALU=1 / SSRLCV_GAUSS_MFMA=1 force
  // one of them for every radius.
  static const bool forceValu = svdev::env("SSRLCV_GAUSS_VALU") != nullptr, forceMfma = svdev::env("SSRLCV_GAUSS_MFMA") != nullptr;
  static const int mfmaMinR = svdev::env("SSRLCV_GAUSS_MFMA_MINR") ? atoi(svdev::env("SSRLCV_GAUSS_MFMA_MINR")) : 11;
  const bool useMfma = forceMfma || (!forceValu && !u8src && RT >= mfmaMinR);
  // small levels (<= 1024^2): the tile kernel (no marching).  Measured inside build_dog on a 4096^2 image (octave 3 =
  // 1024^2, octave 2 = 2048^2): marching kernels everywhere 2.079 ms, tile kernel for octave 3 2.065, for octaves 2 and
  // 3 2.133 (a 2048^2 level is 1024 tiles, four rounds of one-per-CU blocks).  SSRLCV_GAUSS_TILE_MAXPX=<pixels> moves
  // the threshold (0 = never).  Levels with a side below 64 pixels always go here: it is the one kernel that mirrors with
  // the reference's modulo (the marching kernels reflect once, which needs a side of at least 2R).
  static const size_t tileMaxPx = svdev::env("SSRLCV_GAUSS_TILE_MAXPX") ? (size_t)atoll(svdev::env("SSRLCV_GAUSS_TILE_MAXPX")) : ((size_t)1 << 20);
  const bool tiny = w < 64 || h < 64;
  if (tiny || (!u8src && !forceValu && (size_t)w * h <= tileMaxPx)) {
    if (u8src) return SSRLCV_ERR_INVALID_ARG;
    if (canBin) { a.binOut = binOut; if (binned) *binned = true; }
#define SSRLCV_LAUNCH_TILE(RR)                                                                                     \
  do {                                                                                                              \
    static bool attr = false;                                                                                       \
    if (!attr) {                                                                                                    \
      SSRLCV_HIP_TRY(hipFuncSetAttribute((const void*)k_gauss_tile<RR>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                         (int)TileCfg<RR>::ldsBytes));                                              \
      attr = true;                                                                                                  \
    }                                                                                                               \
    hipLaunchKernelGGL(k_gauss_tile<RR>, dim3((w + 63) / 64, (h + 63) / 64), dim3(kTileThreads), TileCfg<RR>::ldsBytes, st, a); \
  } while (0)
    switch (RT) {
      case 6: SSRLCV_LAUNCH_TILE(6); break;
      case 8: SSRLCV_LAUNCH_TILE(8); break;
      case 11: SSRLCV_LAUNCH_TILE(11); break;
      case 16: SSRLCV_LAUNCH_TILE(16); break;
      case 23: SSRLCV_LAUNCH_TILE(23); break;
      default: SSRLCV_LAUNCH_TILE(32); break;
    }
#undef SSRLCV_LAUNCH_TILE
    SSRLCV_LAUNCH_CHECK();
    return SSRLCV_OK;
  }
  // the register-marching kernel (gauss_rm.inc): 256-column strips
  const int rmMask = rm_mask(), rmRows = rm_rows();
  const int R2 = R <= 6 ? 6 : R <= 8 ? 8 : R <= 12 ? 12 : R <= 16 ? 16 : R <= 24 ? 24 : 32;  // radii padded to even values
  const int r2bit = R2 == 6 ? 1 : R2 == 8 ? 2 : R2 == 12 ? 4 : R2 == 16 ? 8 : R2 == 24 ? 16 : 32;
  if ((rmMask & r2bit) && !forceValu && !forceMfma && (size_t)w * h >= rm_min_px() && !u8src && w >= 256 && h >= 64 && (w & 3) == 0 && (h & 3) == 0 && (reinterpret_cast<size_t>(in) & 15) == 0 &&
      (reinterpret_cast<size_t>(out) & 15) == 0 && (uint64_t)w * h * 4 < ((uint64_t)1 << 32)) {
    const uint32_t nS = w / 256, cov = nS * 256;
    ConvArgs r = a;
    if (canBin && cov == w) { r.binOut = binOut; if (binned) *binned = true; }
    memset(r.wgt, 0, sizeof r.wgt);
    for (int k = 0; k <= R; ++k) r.wgt[(R2 - R) + k] = weights_host[k];
#define SSRLCV_LAUNCH_RM_V(RR, OB)                                                                                  \
  do {                                                                                                              \
    constexpr size_t ldsB = OB ? RmCfg<RR>::ldsBytesOneB : RmCfg<RR>::ldsBytes;                                    \
    static int blocksPerCu = 0, cus = 0;                                                                            \
    if (!blocksPerCu) {                                                                                             \
      SSRLCV_HIP_TRY(hipFuncSetAttribute((const void*)k_gauss_rm<RR, OB>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                         (int)ldsB));                                                               \
      int dev = 0, occ = 0;                                                                                         \
      SSRLCV_HIP_TRY(hipGetDevice(&dev));                                                                           \
      SSRLCV_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));                      \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_gauss_rm<RR, OB>, 256, ldsB) != hipSuccess || occ < 1) \
        occ = 1;                                                                                                    \
      blocksPerCu = occ;                                                                                            \
    }                                                                                                               \
    uint32_t by = ((uint32_t)(blocksPerCu * cus) + nS - 1) / nS;                                                    \
    uint32_t rows = (h + by - 1) / by;                                                                              \
    rows = rows < 128 ? 128 : rows;                                                                                 \
    if (rmRows > 0) rows = (uint32_t)rmRows;                                                                        \
    rows = (rows + 15) / 16 * 16;                                                                                   \
    r.rowsPerBlock = rows;                                                                                          \
    hipLaunchKernelGGL((k_gauss_rm<RR, OB>), dim3(nS, (h + rows - 1) / rows), dim3(256), ldsB, st, r);              \
  } while (0)
#define SSRLCV_LAUNCH_RM(RR)                                  \
  do {                                                        \
    if (rm_oneb_mask() & r2bit) SSRLCV_LAUNCH_RM_V(RR, true); \
    else SSRLCV_LAUNCH_RM_V(RR, false);                       \
  } while (0)
    switch (R2) {
      case 6: SSRLCV_LAUNCH_RM(6); break;
      case 8: SSRLCV_LAUNCH_RM(8); break;
      case 12: SSRLCV_LAUNCH_RM(12); break;
      case 16: SSRLCV_LAUNCH_RM(16); break;
      case 24: SSRLCV_LAUNCH_RM(24); break;
      default: SSRLCV_LAUNCH_RM(32); break;
    }
#undef SSRLCV_LAUNCH_RM
#undef SSRLCV_LAUNCH_RM_V
    if (cov < w) launch_valu(a, RT, cov, st);
    SSRLCV_LAUNCH_CHECK();
    return SSRLCV_OK;
  }
  // the MFMA kernel serves full strips of 16-byte aligned rows; what is left of the width goes to the VALU kernel
  // 128-column strips let two blocks (8 waves each) share a CU's 160 KB of LDS (R <= 24).  Alone on the chip they are
  // slower than the 256-column ones (more halo, two accumulation chains per wave instead of four: 0.632 / 0.834 / 0.801
  // against 0.608 / 0.673 / 0.771 ms per 16384^2 level at 23 / 33 / 47 taps), but octaves >= 1 run beside the DoG pass
  // of the previous octave, whose resident blocks keep the big ones off the CUs: the narrow strips are used for the
  // smaller levels (build_dog 2.57 -> 2.48 ms per 4096^2 image).  SSRLCV_GAUSS_WIDE=1 / SSRLCV_GAUSS_NARROW=1 force one.
  static const bool forceWide = svdev::env("SSRLCV_GAUSS_WIDE") != nullptr, forceNarrow = svdev::env("SSRLCV_GAUSS_NARROW") != nullptr;
  const bool wide = R > 24 || forceWide || (!forceNarrow && (size_t)w * h >= ((size_t)1 << 25));
  const uint32_t tw = wide ? 256u : 128u;
  const bool mfmaOk = useMfma && (w & 3) == 0 && (reinterpret_cast<size_t>(in) & 15) == 0 && (reinterpret_cast<size_t>(out) & 15) == 0 &&
                      (uint64_t)w * h * 4 < ((uint64_t)1 << 32) && w >= tw;
  if (!mfmaOk) {
    launch_valu(a, RT, 0, st);
    SSRLCV_LAUNCH_CHECK();
    return SSRLCV_OK;
  }
  const uint32_t nStrips = w / tw, covered = nStrips * tw;
  ConvArgs m = a;
  if (canBin && covered == w) { m.binOut = binOut; if (binned) *binned = true; }
  memset(m.wgt, 0, sizeof m.wgt);
  for (int k = 0; k <= R; ++k) m.wgt[(R2 - R) + k] = weights_host[k];
  // one or two blocks are resident per CU (LDS): size the strips so that the launch is about one round of them;
  // 16-row steps, at least 4 steps of payload per 2R halo
#define SSRLCV_LAUNCH_MFMA2(RR, TW)                                                                                 \
  do {                                                                                                              \
    static int blocksPerCu = 0, cus = 0;                                                                            \
    if (!blocksPerCu) {                                                                                             \
      SSRLCV_HIP_TRY(hipFuncSetAttribute((const void*)k_gauss_mfma2<RR, TW>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                         (int)MfmaCfg<RR, TW>::ldsBytes));                                          \
      int dev = 0, occ = 0;                                                                                         \
      SSRLCV_HIP_TRY(hipGetDevice(&dev));                                                                           \
      SSRLCV_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));                      \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_gauss_mfma2<RR, TW>, kMfmaThreads,       \
                                                       MfmaCfg<RR, TW>::ldsBytes) != hipSuccess || occ < 1)         \
        occ = 1;                                                                                                    \
      blocksPerCu = occ;                                                                                            \
    }                                                                                                               \
    uint32_t by = ((uint32_t)(blocksPerCu * cus) + nStrips - 1) / nStrips;                                          \
    uint32_t rows = (h + by - 1) / by;                                                                              \
    rows = rows < 64 ? 64 : rows;                                                                                   \
    rows = (rows + kMT - 1) / kMT * kMT;                                                                            \
    m.rowsPerBlock = rows;                                                                                          \
    const size_t ldsB2 = MfmaCfg<RR, TW>::ldsBytes;                                                                 \
    hipLaunchKernelGGL((k_gauss_mfma2<RR, TW>), dim3(nStrips, (h + rows - 1) / rows), dim3(kMfmaThreads), ldsB2, st, m); \
  } while (0)
  switch (R2) {
    case 6: if (wide) SSRLCV_LAUNCH_MFMA2(6, 256); else SSRLCV_LAUNCH_MFMA2(6, 128); break;
    case 8: if (wide) SSRLCV_LAUNCH_MFMA2(8, 256); else SSRLCV_LAUNCH_MFMA2(8, 128); break;
    case 12: if (wide) SSRLCV_LAUNCH_MFMA2(12, 256); else SSRLCV_LAUNCH_MFMA2(12, 128); break;
    case 16: if (wide) SSRLCV_LAUNCH_MFMA2(16, 256); else SSRLCV_LAUNCH_MFMA2(16, 128); break;
    case 24: if (wide) SSRLCV_LAUNCH_MFMA2(24, 256); else SSRLCV_LAUNCH_MFMA2(24, 128); break;
    default: SSRLCV_LAUNCH_MFMA2(32, 256); break;
  }
#undef SSRLCV_LAUNCH_MFMA2
  if (covered < w) launch_valu(a, RT, covered, st);  // the partial last strip
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

namespace svp {
int stream_priority_mode() {
  static const int m = svdev::env("SSRLCV_PRIO") ? atoi(svdev::env("SSRLCV_PRIO")) : 0;
  return m;
}
PlanAsync* plan_async(const ssrlcv_sift_plan* plan) {
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (plan->asyncState == 0) {
    plan->asyncState = -1;
    if (!svdev::env("SSRLCV_SIFT_SERIAL")) {
      PlanAsync* a = new (std::nothrow) PlanAsync();
      bool ok = a != nullptr;
      auto mk = [&](hipEvent_t& e) { ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess; };
      if (ok) {
        // per-device side streams, created once and kept for the life of the process
        struct Side { hipStream_t chain, table, chain2, polar; };
        static std::map<int, Side> pool;
        int dev = 0;
        ok = hipGetDevice(&dev) == hipSuccess;
        auto it = pool.find(dev);
        if (ok && it == pool.end()) {
          Side pr{nullptr, nullptr, nullptr, nullptr};
          int least = 0, greatest = 0;
          (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
          const int tablePrio = svdev::env("SSRLCV_SIDE_LOW_PRIORITY") ? least : 0;
          // SSRLCV_PRIO (developer build): 1 = the gradient tables of describe on their own LOW-priority stream, so that the list
          // chains' short launches get the CU slots the tables' thousands of short-lived blocks keep freeing; 2 = and the
          // chains' side streams at HIGH priority (see svp::stream_priority_mode)
          const int prioMode = svp::stream_priority_mode();
          ok = hipStreamCreateWithPriority(&pr.chain, hipStreamNonBlocking, (prioMode & 2) ? greatest : 0) == hipSuccess &&
               hipStreamCreateWithPriority(&pr.table, hipStreamNonBlocking, tablePrio) == hipSuccess &&
               hipStreamCreateWithPriority(&pr.chain2, hipStreamNonBlocking, (prioMode & 2) ? greatest : 0) == hipSuccess &&
               hipStreamCreateWithPriority(&pr.polar, hipStreamNonBlocking, (prioMode & 1) ? least : 0) == hipSuccess;
          if (ok) it = pool.emplace(dev, pr).first;
        }
        if (ok) {
          a->chain = it->second.chain;
          a->table = it->second.table;
          a->chain2 = it->second.chain2;
          a->polar = it->second.polar;
        }
        mk(a->fork);
        for (hipEvent_t& e : a->join) mk(e);
        for (hipEvent_t& e : a->convDone) mk(e);
        for (hipEvent_t& e : a->dogDone) mk(e);
        for (hipEvent_t& e : a->polarDone) mk(e);
        for (hipEvent_t& e : a->binDone) mk(e);
        mk(a->groupFork);
        for (hipEvent_t& e : a->groupExpanded) mk(e);
        for (hipEvent_t& e : a->groupReady) mk(e);
        mk(a->expandFork);
        for (hipEvent_t& e : a->expandJoin) mk(e);
        for (auto& lv : a->levelDone)
          for (hipEvent_t& e : lv) mk(e);
      }
      if (ok) {
        plan->async = a;
        plan->asyncState = 1;
      } else {
        delete a;  // a partially created set leaks its handles; creation only fails when the device is unusable
        (void)hipGetLastError();
      }
    }
  }
  return plan->asyncState == 1 ? plan->async : nullptr;
}
}  // namespace svp

extern "C" {

int ssrlcv_gauss_kernel_host(float sigma, float pixelWidth, float* weights) {
  // Blur::Blur (src/FeatureFactory.cu:15-18,29-33); PI is the double macro of include/common_includes.hpp:46
  int ksize = (int)ceilf((float)8 * sigma / pixelWidth);
  if (ksize % 2 == 0) ksize++;
  if (ksize > svp::kMaxTaps) return -ksize;
  int i = 0;
  for (int x = -ksize / 2; x <= ksize / 2; ++x, ++i) {
    weights[i] = expf(-((x * x) / 2.0f / sigma / sigma)) / sqrtf((float)(2.0f * SSRLCV_PI_D)) / sigma;
  }
  return ksize;
}

int ssrlcv_hip_convert_to_bw(const uint8_t* colorPixels, uint32_t colorDepth, uint8_t* bw, size_t numPixels,
                             ssrlcv_stream_t stream) {
  if (!colorPixels || !bw || colorDepth < 2 || colorDepth > 4) return SSRLCV_ERR_INVALID_ARG;
  if (numPixels == 0) return SSRLCV_OK;
  hipLaunchKernelGGL(k_to_bw, dim3((unsigned)((numPixels + 255) / 256)), dim3(256), 0, (hipStream_t)stream, colorPixels,
                     (int)colorDepth, bw, numPixels);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

int ssrlcv_hip_u8_to_f32(const uint8_t* pixels, float* out, size_t n, ssrlcv_stream_t stream) {
  if (!pixels || !out) return SSRLCV_ERR_INVALID_ARG;
  if (n == 0) return SSRLCV_OK;
  hipLaunchKernelGGL(k_u8_to_f32, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, pixels, out, n);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

int ssrlcv_hip_upsample2x(const float* in, uint32_t w, uint32_t h, float* out, ssrlcv_stream_t stream) {
  if (!in || !out || !w || !h) return SSRLCV_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_upsample2x<float>, dim3((2 * w + 255) / 256, 2 * h), dim3(256), 0, (hipStream_t)stream, in, w, h, out);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

int ssrlcv_hip_upsample2x_u8(const uint8_t* in, uint32_t w, uint32_t h, float* out, ssrlcv_stream_t stream) {
  if (!in || !out || !w || !h) return SSRLCV_ERR_INVALID_ARG;
  if ((w & 1) == 0 && (reinterpret_cast<size_t>(out) & 15) == 0)
    hipLaunchKernelGGL(k_upsample2x_u8x4, dim3((2 * w / 4 + 255) / 256, 2 * h), dim3(256), 0, (hipStream_t)stream, in, w,
                       h, out);
  else
    hipLaunchKernelGGL(k_upsample2x<uint8_t>, dim3((2 * w + 255) / 256, 2 * h), dim3(256), 0, (hipStream_t)stream, in, w,
                       h, out);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

int ssrlcv_hip_bin2x(const float* in, uint32_t w, uint32_t h, float* out, ssrlcv_stream_t stream) {
  if (!in || !out || w < 2 || h < 2 || (w & 1)) return SSRLCV_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_bin2x, dim3((w / 2 + 255) / 256, h / 2), dim3(256), 0, (hipStream_t)stream, in, w, h, out);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

int ssrlcv_hip_gauss_sep_conv(const float* in, float* out, float* tmp, uint32_t w, uint32_t h, int taps,
                              const float* weights_host, float* minmax, ssrlcv_stream_t stream) {
  if (!in || !out || !weights_host || !w || !h || in == out) return SSRLCV_ERR_INVALID_ARG;
  return launch_conv(in, out, tmp, w, h, taps, weights_host, minmax, (hipStream_t)stream);
}

int ssrlcv_hip_minmax(const float* in, size_t n, float* minmax, ssrlcv_stream_t stream) {
  if (!in || !minmax || !n) return SSRLCV_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_init_minmax, dim3(1), dim3(64), 0, st, minmax, 1);
  size_t blocks = (n + 1023) / 1024;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(k_minmax, dim3((unsigned)blocks), dim3(256), 0, st, in, n, minmax);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

int ssrlcv_hip_normalize(float* data, size_t n, const float* minmax, ssrlcv_stream_t stream) {
  if (!data || !minmax || !n) return SSRLCV_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_normalize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, data, n, minmax);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

// partial (nullable): workspace for the atomic-free mode (float[2 * kDog][kDogMaxWaves]); with it `maxBlocks` may be large
// (short-lived blocks), without it the grid stays small because every block ends with same-address atomics
static int launch_dog(const float* const levels_host[6], const float* levelMinMax, uint32_t w, uint32_t h,
                      float* const dog_host[5], float* dogMinMax, int firstDog, int lastDog, unsigned maxBlocks,
                      float* partial, ssrlcv_stream_t stream) {
  if (!levels_host || !levelMinMax || !dog_host || !w || !h) return SSRLCV_ERR_INVALID_ARG;
  if (firstDog < 0 || lastDog > svp::kDog || firstDog >= lastDog) return SSRLCV_ERR_INVALID_ARG;
  size_t n = (size_t)w * h;
  if (n % 4) return SSRLCV_ERR_INVALID_ARG;
  DogArgs a;
  for (int b = 0; b < svp::kGauss; ++b) a.lvl[b] = levels_host[b];
  for (int b = 0; b < svp::kDog; ++b) a.dog[b] = dog_host[b];
  a.lvlMinMax = levelMinMax;
  a.dogMinMax = dogMinMax;
  a.partial = partial;
  a.n = n;
  size_t blocks = (n / 4 + 255) / 256;
  if (partial && dogMinMax) {
    if (maxBlocks > svp::kDogMaxBlocks) maxBlocks = svp::kDogMaxBlocks;
    if (blocks > maxBlocks) blocks = maxBlocks;
    hipLaunchKernelGGL(k_dog<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, firstDog, lastDog);
    hipLaunchKernelGGL(k_dog_finalize, dim3((unsigned)(lastDog - firstDog)), dim3(kFinalizeThreads), 0, (hipStream_t)stream, partial,
                       (unsigned)blocks * 4, firstDog, dogMinMax);
  } else {
    if (blocks > 1024) blocks = 1024;  // 4 blocks per CU; each block ends with up to 10 same-address atomics
    hipLaunchKernelGGL(k_dog<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, firstDog, lastDog);
  }
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}
int ssrlcv_hip_dog_normalised_sub(const float* const levels_host[6], const float* levelMinMax, uint32_t w, uint32_t h,
                                  float* const dog_host[5], float* dogMinMax, ssrlcv_stream_t stream) {
  return launch_dog(levels_host, levelMinMax, w, h, dog_host, dogMinMax, 0, svp::kDog, 1024, nullptr, stream);
}

// ---- plan ------------------------------------------------------------------------------------------------------------
int ssrlcv_sift_plan_create(uint32_t w, uint32_t h, const ssrlcv_sift_params* params, ssrlcv_sift_plan** out) {
  if (!out || !params || !w || !h) return SSRLCV_ERR_INVALID_ARG;
  if (params->maxOrientations == 0 || params->maxOrientations > (uint32_t)svp::kMaxOrient) return SSRLCV_ERR_UNSUPPORTED;
  // The sampling kernels index their windows with 16-bit multiply-high divisions: the descriptor window half-width
  // ceil(lambda * sigma / pixelWidth) must stay below 256 and the orientation window 2 ceil(3 lambda_o sigma /
  // pixelWidth) + 1 below 256.  sigma / pixelWidth of a refined key point is below 0.5 sqrt(2) * sqrt(2)^5 / 0.5 = 8,
  // so the reference's 6.0 / 1.5 are far inside; absurd widths are refused instead of sampled wrongly.
  if (!(params->descriptorContribWidth > 0.0f) || params->descriptorContribWidth > 30.0f ||
      !(params->orientationContribWidth > 0.0f) || params->orientationContribWidth > 5.0f)
    return SSRLCV_ERR_UNSUPPORTED;
  // "image too small" check of ScaleSpace::ScaleSpace (src/FeatureFactory.cu:341-345): numResize = 2^(start+depth.x)
  if (w / 8 == 0 || h / 8 == 0) return SSRLCV_ERR_INVALID_ARG;
  // S3, makeBinnable as ScaleSpace::ScaleSpace applies it (src/FeatureFactory.cu:364-376, src/Image.cu:966-995): even
  // sizes are padded to multiples of 2^(startingOctave + depth.x) = 8 before the upsample, sizes with an odd side to
  // multiples of 2^(depth.x + 1) = 32 after it; border = (numResize - size % numResize) / 2 on both sides, zero filled
  int padMode = 0;
  uint32_t padX = 0, padY = 0, ow = w * 2, oh = h * 2;
  if (w % 2 == 0 && h % 2 == 0) {
    if (w % 8 || h % 8) {
      padMode = 1;
      padX = w % 8 ? (8 - w % 8) / 2 : 0;
      padY = h % 8 ? (8 - h % 8) / 2 : 0;
      ow = (w + 2 * padX) * 2;
      oh = (h + 2 * padY) * 2;
    }
  } else if (ow % 32 || oh % 32) {
    padMode = 2;
    padX = ow % 32 ? (32 - ow % 32) / 2 : 0;
    padY = oh % 32 ? (32 - oh % 32) / 2 : 0;
    ow += 2 * padX;
    oh += 2 * padY;
  }
  // The smallest octave is an eighth of octave 0.  Below 16 pixels on a side the 65-tap mirror of the reference
  // (getSymmetrizedCoord, src/Image.cu:1248-1252: `(i + 2l) % 2l` with i + 2l < 0) indexes outside the level, i.e. upstream
  // itself is undefined for inputs under 64 pixels: refused here instead of inventing a value.
  if (ow < 128 || oh < 128) return SSRLCV_ERR_UNSUPPORTED;
  // one polar table (8 bytes per pixel of octave 0) is addressed with 32-bit byte offsets by the orientation kernel
  if (svp::polar_level_stride(ow, oh) * 8 >= ((size_t)1 << 32)) return SSRLCV_ERR_UNSUPPORTED;
  ssrlcv_sift_plan* p = new (std::nothrow) ssrlcv_sift_plan;
  if (!p) return SSRLCV_ERR_INVALID_ARG;
  memset(p, 0, sizeof *p);
  p->W = w;
  p->H = h;
  p->padMode = padMode;
  p->padX = padX;
  p->padY = padY;
  p->params = *params;
  p->stopStage = 7;
  p->fusedCall = 0;
  p->polarInFlight = 0;
  p->chain0InFlight = 0;
  p->stageEvent = nullptr;
  p->async = nullptr;
  p->asyncState = 0;
  // sigma ladder: SIFT_FeatureFactory.cu:63-64 + FeatureFactory.cu:383-399
  float sigmas[svp::kGauss];
  float mulY = sqrtf(2.0f), mulX = 2;
  sigmas[0] = sqrtf(2.0f) / 2.0f;
  for (int i = 1; i < svp::kGauss; ++i) sigmas[i] = sigmas[i - 1] * mulY;
  float pixelWidth = 1.0f;
  pixelWidth /= 2.0f;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t r = off; off += align256(bytes); return r; };
  size_t P0 = (size_t)ow * oh;
  p->off_pad = take(padMode == 1 ? (size_t)(w + 2 * padX) * (h + 2 * padY) : padMode == 2 ? (size_t)w * h * 16 : 0);
  p->off_in0 = take(P0 * 4);
  p->off_in1 = take(P0 / 4 * 4);
  p->off_in2 = take(P0 / 16 * 4);
  for (int o = 0; o < svp::kOctaves; ++o)
    for (int b = 0; b < svp::kGauss; ++b) p->off_gauss[o][b] = take((P0 >> (2 * o)) * 4);  // P_o = P_0 / 4^o floats
  p->off_minmax = take(sizeof(float) * 2 * (svp::kGauss + svp::kDog) * svp::kOctaves);
  p->off_state = take(sizeof(svp::OctaveState) * svp::kOctaves);
  uint32_t maxFeat = 0;
  for (int o = 0; o < svp::kOctaves; ++o) {
    svp::OctavePlan& oc = p->oct[o];
    oc.w = ow;
    oc.h = oh;
    oc.pixelWidth = pixelWidth;
    for (int b = 0; b < svp::kGauss; ++b) {
      oc.sigma[b] = sigmas[b];
      oc.taps[b] = ssrlcv_gauss_kernel_host(sigmas[b], pixelWidth, oc.weights[b]);
      if (oc.taps[b] <= 0 || oc.taps[b] > 65) { delete p; return SSRLCV_ERR_UNSUPPORTED; }
    }
    size_t P = (size_t)ow * oh;
    oc.off_flags = take(P);
    oc.off_polar = take(svp::polar_level_stride(oc.w, oc.h) * 3 * 8);
    uint32_t cap = params->maxKeyPointsPerOctave ? params->maxKeyPointsPerOctave : (uint32_t)(P / 16);
    if (cap < 4096) cap = 4096;
    oc.cap = cap;
    oc.off_kpA = take((size_t)cap * sizeof(ssrlcv_sskeypoint));
    oc.off_kpB = take((size_t)cap * sizeof(ssrlcv_sskeypoint));
    oc.off_theta = take((size_t)cap * svp::kMaxOrient * sizeof(float));
    oc.off_thetaCnt = take((size_t)cap * sizeof(uint32_t));
    // partition scratch: pixel-domain (3 keys over P) is the largest user
    // (compact.h: NKEYS x 4 wave runs per 256-thread chunk; chunks of 256 x 32 pixels / 256 x 8 list entries)
    size_t words = (size_t)3 * 4 * ((P + 8191) / 8192) + 16 +
                   (size_t)svp::kDog * 4 * (((size_t)cap * svp::kMaxOrient + 2047) / 2048) + 16;
    oc.off_part = take(words * 4);
    oc.off_featBase = take(256);
    oc.off_descConst = take((size_t)cap * 32);
    maxFeat += cap;
    if (o + 1 < svp::kOctaves) {
      ow /= 2;
      oh /= 2;
      pixelWidth *= 2.0f;
      for (int b = 0; b < svp::kGauss; ++b) sigmas[b] *= mulX;
    }
  }
  p->off_extremaCounts = take(256);
  p->off_dogPartial = take(sizeof(float) * 2 * svp::kDog * svp::kDogMaxWaves * svp::kOctaves);  // one region per octave: their DoG kernels may run side by side
  p->off_groups = take(4096);
  p->total = off;
  p->maxFeatures = maxFeat;
  *out = p;
  return SSRLCV_OK;
}

void ssrlcv_sift_plan_destroy(ssrlcv_sift_plan* plan) {
  if (!plan) return;
  if (plan->async) {
    svp::PlanAsync* a = plan->async;
    (void)hipEventDestroy(a->fork);
    for (hipEvent_t e : a->join) (void)hipEventDestroy(e);
    for (hipEvent_t e : a->convDone) (void)hipEventDestroy(e);
    for (hipEvent_t e : a->dogDone) (void)hipEventDestroy(e);
    for (hipEvent_t e : a->polarDone) (void)hipEventDestroy(e);
    for (hipEvent_t e : a->binDone) (void)hipEventDestroy(e);
    (void)hipEventDestroy(a->expandFork);
    (void)hipEventDestroy(a->groupFork);
    for (hipEvent_t e : a->groupExpanded) (void)hipEventDestroy(e);
    for (hipEvent_t e : a->groupReady) (void)hipEventDestroy(e);
    for (hipEvent_t e : a->expandJoin) (void)hipEventDestroy(e);
    for (auto& lv : a->levelDone)
      for (hipEvent_t e : lv) (void)hipEventDestroy(e);
    delete a;
  }
  delete plan;
}
size_t ssrlcv_sift_plan_workspace_bytes(const ssrlcv_sift_plan* plan) { return plan ? plan->total : 0; }
uint32_t ssrlcv_sift_plan_max_features(const ssrlcv_sift_plan* plan) { return plan ? plan->maxFeatures : 0; }
void ssrlcv_sift_plan_set_stop_stage(ssrlcv_sift_plan* plan, int stage) { if (plan) plan->stopStage = stage; }
int ssrlcv_sift_plan_set_stage_event(ssrlcv_sift_plan* plan, void* event) {
  if (!plan) return SSRLCV_ERR_INVALID_ARG;
  plan->stageEvent = (hipEvent_t)event;
  return SSRLCV_OK;
}

int ssrlcv_sift_plan_level(const ssrlcv_sift_plan* plan, void* workspace, int kind, int octave, int blur, float** data,
                           uint32_t* w, uint32_t* h, float** minmax_dev) {
  if (!plan || !workspace || octave < 0 || octave >= svp::kOctaves) return SSRLCV_ERR_INVALID_ARG;
  char* ws = (char*)workspace;
  const svp::OctavePlan& oc = plan->oct[octave];
  float* mm = (float*)(ws + plan->off_minmax) + (size_t)octave * 2 * (svp::kGauss + svp::kDog);
  if (kind == 0) {
    // The pipeline never writes a DoG level (see k_dogx).  For inspection the requested level is materialised here, by
    // the kernel-level DoG entry point's kernel, from the two Gaussian levels the workspace keeps, into the scratch region
    // that holds octave 0's input during build_dog; its {min, max} are the ones build_dog reduced.  Synchronous; the
    // pointer is valid until the next call on this workspace.
    if (blur < 0 || blur >= svp::kDog) return SSRLCV_ERR_INVALID_ARG;
    if (((size_t)oc.w * oc.h) % 4) return SSRLCV_ERR_UNSUPPORTED;
    SSRLCV_HIP_TRY(hipDeviceSynchronize());
    const float* lv[svp::kGauss];
    float* dogs[svp::kDog];
    float* scratch = (float*)(ws + plan->off_in0);
    for (int b = 0; b < svp::kGauss; ++b) lv[b] = (const float*)(ws + plan->off_gauss[octave][b]);
    for (int b = 0; b < svp::kDog; ++b) dogs[b] = scratch;
    int rc = launch_dog(lv, mm, oc.w, oc.h, dogs, nullptr, blur, blur + 1, 1024, nullptr, nullptr);
    if (rc) return rc;
    SSRLCV_HIP_TRY(hipDeviceSynchronize());
    if (data) *data = scratch;
    if (minmax_dev) *minmax_dev = mm + 2 * (svp::kGauss + blur);
  } else if (kind == 1) {
    if (blur < 0 || blur >= svp::kGauss) return SSRLCV_ERR_INVALID_ARG;
    if (data) *data = (float*)(ws + plan->off_gauss[octave][blur]);
    if (minmax_dev) *minmax_dev = mm + 2 * blur;
  } else {
    return SSRLCV_ERR_INVALID_ARG;
  }
  if (w) *w = oc.w;
  if (h) *h = oc.h;
  return SSRLCV_OK;
}

// ScaleSpace::ScaleSpace with makeDOG = true (src/FeatureFactory.cu:338-440) + searchForExtrema's findExtrema (:847-882)
namespace {
// Schedule of the fused DoG / extrema pass (k_dogx).  It needs the levels complete (global min / max), so it follows the
// convolutions of its octave, on a side stream beside the convolutions of the next octave.  A split form exists: the part
// that needs only levels 0..3 (DoG 0..2: their min / max and the extrema of DoG level 1) starts as soon as gaussian level 3
// is complete, beside the convolutions of levels 4 and 5; the rest (DoG 1..4 from levels 1..5: extrema of levels 2 and 3,
// min / max of 3 and 4) follows after level 5 and ORs its bits into the flag bytes.  With the materialising DoG kernel of
// round 2 (HBM bound) the split paid on the 2^26-pixel octave; the fused pass is VALU bound and reads levels 1..3 twice in
// the split form: measured on a 4096^2 image 2.14 ms split on octave 0 against 1.96 ms whole, so the default is whole.
// SSRLCV_DOG_SPLIT=1: split on every octave (0: never).  SSRLCV_DOGX_WAVES: waves per launch (default 14336: build_dog of a
// 4096^2 image, median of 30, two sweeps: 8192 waves 1.760-1.776 ms, 10240 1.73-1.75, 12288 1.70-1.71, 14336 1.698-1.699,
// 16384 1.70-1.71, 20480 1.72-1.73, 24576 1.71-1.72).
struct DogSchedule {
  int split;  // 0 never (default), 1 always
  unsigned waves;
  DogSchedule() {
    split = 0;
    if (const char* e = svdev::env("SSRLCV_DOG_SPLIT")) split = atoi(e) != 0 ? 1 : 0;
    waves = 14336;
    if (const char* e = svdev::env("SSRLCV_DOGX_WAVES")) waves = (unsigned)atoi(e) > 0 ? (unsigned)atoi(e) : 14336;
  }
};
}  // namespace

int ssrlcv_hip_sift_build_dog(const ssrlcv_sift_plan* plan, const uint8_t* pixels, void* workspace,
                              ssrlcv_stream_t stream) {
  if (!plan || !pixels || !workspace) return SSRLCV_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  float* mmAll = (float*)(ws + plan->off_minmax);
  const int pairs = (svp::kGauss + svp::kDog) * svp::kOctaves;
  // The convolutions (FMA-issue bound) stay on the caller's stream; the DoG kernel of octave o (HBM bound) runs on a
  // side stream beside the convolutions of octave o+1, which only need level 3 of octave o.  Octaves alternate between
  // two sets of gaussian buffers so that octave o+1 never overwrites what DoG(o) is still reading.
  svp::PlanAsync* as = svp::plan_async(plan);
  hipStream_t sd = as ? as->table : st;
  static const DogSchedule sched;
  static const bool overlapOctaves = svdev::env("SSRLCV_NO_OCTAVE_OVERLAP") == nullptr;
  static const int overlapFrom = svdev::env("SSRLCV_OCTAVE_OVERLAP_FROM") ? atoi(svdev::env("SSRLCV_OCTAVE_OVERLAP_FROM")) : 1;
  static const int deferDog0 = [] {  // 1 or 2 (octave 3 has no successor to wait for), 0: off
    const int n = svdev::env("SSRLCV_DOGX0_AFTER") ? atoi(svdev::env("SSRLCV_DOGX0_AFTER")) : 0;
    return n == 1 || n == 2 ? n : 0;
  }();
  // Developer build, SSRLCV_EARLY_POLAR=1: a fused extract starts an octave's gradient tables (k_polar, the first thing the
  // key-point stage needs) on a side stream as soon as that octave's DoG pass is through, beside the small octaves' launches.
  // Exact, measured in round 5 and NOT the default: the scale-space stage grows by 0.36 ms per 4096^2 image (the tables'
  // 16 384 short blocks crowd the small octaves' latency-bound launches) and the key-point stage shrinks by 0.10
  // (profiles/r05_schedule_ab.txt: step 10.19 -> 10.67 ms).
  static const bool wantEarlyPolar = svdev::env("SSRLCV_EARLY_POLAR") != nullptr;
  const bool earlyPolar = as && plan->fusedCall && plan->stopStage >= 6 && wantEarlyPolar;
  plan->polarInFlight = 0;
  plan->chain0InFlight = 0;  // (a describe that failed half-way may have left it set)
  if (earlyPolar) {  // the tables may still be read by the sampling kernels of the previous extract on this plan: behind the caller's stream
    SSRLCV_HIP_TRY(hipEventRecord(as->fork, st));
    SSRLCV_HIP_TRY(hipStreamWaitEvent(as->polar, as->fork, 0));
  }
  hipLaunchKernelGGL(k_init_minmax, dim3(1), dim3(64), 0, st, mmAll, pairs);
  // S1+S2: u8 -> f32 + one 2x upsample (startingOctave = -1)
  float* in = (float*)(ws + plan->off_in0);
  int rc;
  bool fuseUpsample = false;
  if (plan->padMode == 1) {  // S3 before S2: zero border around the u8 input ((float)0 == 0.0f: the order of S1 and S3 is free)
    const uint32_t pw = plan->W + 2 * plan->padX, ph = plan->H + 2 * plan->padY;
    uint8_t* padded = (uint8_t*)(ws + plan->off_pad);
    hipLaunchKernelGGL(k_add_border<uint8_t>, dim3((pw + 255) / 256, ph), dim3(256), 0, st, pixels, plan->W, plan->H, padded,
                       plan->padX, plan->padY);
    rc = ssrlcv_hip_upsample2x_u8(padded, pw, ph, in, stream);
  } else if (plan->padMode == 2) {  // S2 first, then the border around the upsampled image
    float* up = (float*)(ws + plan->off_pad);
    rc = ssrlcv_hip_upsample2x_u8(pixels, plan->W, plan->H, up, stream);
    hipLaunchKernelGGL(k_add_border<float>, dim3((plan->oct[0].w + 255) / 256, plan->oct[0].h), dim3(256), 0, st, up,
                       2 * plan->W, 2 * plan->H, in, plan->padX, plan->padY);
  } else if (upsample_fusable(plan->oct[0].w, plan->oct[0].h, plan->oct[0].taps[0])) {
    rc = 0;  // S1 + S2 happen in the loader of octave 0's first level
    fuseUpsample = true;
  } else {
    rc = ssrlcv_hip_upsample2x_u8(pixels, plan->W, plan->H, in, stream);
  }
  if (rc) return rc;
  float* nextIn[3] = {(float*)(ws + plan->off_in1), (float*)(ws + plan->off_in2), (float*)(ws + plan->off_in1)};
  const float firstNoise = (float)(svp::kNoiseThreshold * 0.8);  // removeNoise(noiseThreshold * 0.8), src/FeatureFactory.cu:484
  // Experimental schedule (developer build, SSRLCV_PHASED=<n + 1>): the dependency chain of the stage -- levels 0-3 of every
  // octave, each octave's input being the bin of the previous one's level 3 -- runs in order on the caller's stream (phases
  // A_0 .. A_3); what hangs off it (levels 4-5 and the DoG / extrema pass of octave o: phase B_o) goes to side streams and
  // may be held back until the chain is n octaves ahead (B_o waits for the bin of octave o + n), so that the long-lived
  // blocks of the big levels do not sit on the CUs while the chain's short launches look for room.
  static const int phased = svdev::env("SSRLCV_PHASED") ? atoi(svdev::env("SSRLCV_PHASED")) : 0;
  if (as && phased > 0) {
    const int ahead = phased - 1;
    const float* octIn = in;
    for (int o = 0; o < svp::kOctaves; ++o) {  // the chain
      const svp::OctavePlan& oc = plan->oct[o];
      float* mm = mmAll + (size_t)o * 2 * (svp::kGauss + svp::kDog);
      const float* src = octIn;
      for (int b = 0; b < 4; ++b) {
        float* dst = (float*)(ws + plan->off_gauss[o][b]);
        bool binned = false;
        rc = launch_conv(src, dst, nullptr, oc.w, oc.h, oc.taps[b], oc.weights[b], mm + 2 * b, st,
                         (fuseUpsample && o == 0 && b == 0) ? pixels : nullptr, (b == 3 && o + 1 < svp::kOctaves) ? nextIn[o] : nullptr, &binned);
        if (rc) return rc;
        src = dst;
        if (b == 3 && o + 1 < svp::kOctaves) {
          if (!binned) {
            rc = ssrlcv_hip_bin2x(dst, oc.w, oc.h, nextIn[o], (ssrlcv_stream_t)st);
            if (rc) return rc;
          }
          octIn = nextIn[o];
        }
      }
      SSRLCV_HIP_TRY(hipEventRecord(as->binDone[o], st));
    }
    for (int o = 0; o < svp::kOctaves; ++o) {  // what hangs off it
      const svp::OctavePlan& oc = plan->oct[o];
      float* mm = mmAll + (size_t)o * 2 * (svp::kGauss + svp::kDog);
      const hipStream_t sb = o == 0 ? as->table : (o == 1 ? as->chain : st);
      const int gate = o + ahead < svp::kOctaves ? o + ahead : svp::kOctaves - 1;
      if (sb != st) SSRLCV_HIP_TRY(hipStreamWaitEvent(sb, as->binDone[gate], 0));
      const float* lv[svp::kGauss];
      for (int b = 0; b < svp::kGauss; ++b) lv[b] = (const float*)(ws + plan->off_gauss[o][b]);
      for (int b = 4; b < svp::kGauss; ++b) {
        rc = launch_conv(lv[b - 1], (float*)(ws + plan->off_gauss[o][b]), nullptr, oc.w, oc.h, oc.taps[b], oc.weights[b], mm + 2 * b, sb);
        if (rc) return rc;
      }
      float* dogPartial = (float*)(ws + plan->off_dogPartial) + (size_t)o * 2 * svp::kDog * svp::kDogMaxWaves;
      rc = launch_dogx(lv, mm, oc.w, oc.h, (uint8_t*)(ws + oc.off_flags), mm + 2 * svp::kGauss, dogPartial, 0, svp::kDog, 0, false, firstNoise,
                       sched.waves, sb);
      if (rc) return rc;
      SSRLCV_HIP_TRY(hipEventRecord(as->dogDone[o], sb));
      if (earlyPolar) {
        SSRLCV_HIP_TRY(hipStreamWaitEvent(as->polar, as->dogDone[o], 0));
        svp::launch_polar_octave(plan, ws, o, as->polar);
        SSRLCV_HIP_TRY(hipEventRecord(as->polarDone[o], as->polar));
      }
    }
    if (earlyPolar) plan->polarInFlight = 1;
    SSRLCV_HIP_TRY(hipStreamWaitEvent(st, as->dogDone[0], 0));
    SSRLCV_HIP_TRY(hipStreamWaitEvent(st, as->dogDone[1], 0));
    return SSRLCV_OK;
  }
  for (int o = 0; o < svp::kOctaves; ++o) {
    const svp::OctavePlan& oc = plan->oct[o];
    float* mm = mmAll + (size_t)o * 2 * (svp::kGauss + svp::kDog);
    const size_t* offGauss = plan->off_gauss[o];
    // An octave only needs level 3 of the one before (its bin), so octaves alternate between the caller's stream and a side
    // stream from octave `overlapFrom` on: the next octave's first levels run beside levels 4-5 of the current one.
    // Rounds 2-3 started at octave 2 (levels 4-5 of octave 0 were MFMA blocks that filled every CU's LDS: octave 1 beside
    // them only waited); with level 4 on the register-marching kernel octave 1 finds room: end of round 4, the bench step
    // 10.13 -> 10.09 ms, the stage 1.708 -> 1.683 ms per 4096^2 image (two runs each), so the default is 1.
    const hipStream_t so = (as && overlapOctaves && o >= overlapFrom && ((o - overlapFrom) & 1) == 0) ? as->chain : st;
    // DoG / extrema passes: octaves 0 and 1 on the side stream (beside the next octave's convolutions); with the overlap
    // the last two follow their own convolutions on those streams -- behind octave 1's in one in-order stream they were the tail
    const hipStream_t sdo = (as && overlapOctaves && o >= 2) ? so : ((deferDog0 && as && o == 1) ? as->chain2 : sd);  // (o >= 2 whatever the first overlapped octave)
    float* dogPartial = (float*)(ws + plan->off_dogPartial) + (size_t)o * 2 * svp::kDog * svp::kDogMaxWaves;
    uint8_t* flags = (uint8_t*)(ws + oc.off_flags);
    if (as && o >= 1) SSRLCV_HIP_TRY(hipStreamWaitEvent(so, as->binDone[o - 1], 0));
    const float* src = in;
    const float* lv[svp::kGauss] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int b = 0; b < svp::kGauss; ++b) lv[b] = (const float*)(ws + offGauss[b]);
    const bool split = as && sched.split == 1;
    // (Round 3 handed the row chunks of odd levels out bottom-up, so that a level starts on the rows its predecessor wrote
    // last and finds them in the 256 MB memory-side cache: build_dog 2.07-2.11 ms against 2.05-2.08 top-down, no gain.)
    // Levels 0 + 1 in one launch (gauss_pair_rm.inc) where that wins: the u8-sourced first octave from 2^26 pixels up (per
    // 8192^2 level pair 220 us against 253 for the two launches; a 4096^2 pair 77 against 68, and with a float source the
    // fused form only breaks even at 8192^2: tools/gauss_pair_lab.hip, profiles/r06_kernel_ab.txt).
    // SSRLCV_GAUSS_PAIR_MINPX=<pixels> moves the threshold (developer build), SSRLCV_NO_GAUSS_PAIR=1 turns it off.
    static const size_t pairMinPx = svdev::env("SSRLCV_GAUSS_PAIR_MINPX") ? (size_t)atoll(svdev::env("SSRLCV_GAUSS_PAIR_MINPX")) : ((size_t)1 << 26);
    // float-sourced octaves (o >= 1): SSRLCV_GAUSS_PAIR_MINPX_F32=<pixels>
    static const size_t pairMinPxF32 = svdev::env("SSRLCV_GAUSS_PAIR_MINPX_F32") ? (size_t)atoll(svdev::env("SSRLCV_GAUSS_PAIR_MINPX_F32")) : SSRLCV_PAIR_F32_MINPX;
    int firstLevel = 0;
    const bool u8Pair = o == 0 && fuseUpsample && (size_t)oc.w * oc.h >= pairMinPx;
    const bool f32Pair = !(o == 0 && fuseUpsample) && (size_t)oc.w * oc.h >= pairMinPxF32;
    if ((u8Pair || f32Pair) && pair_rm_usable(oc.w, oc.h) &&
        pair_usable(oc.w, oc.h, oc.taps[0], oc.taps[1], f32Pair ? in : (const float*)(ws + offGauss[0]), (float*)(ws + offGauss[0]), (float*)(ws + offGauss[1]))) {
      // (developer build, SSRLCV_GAUSS_PAIR_FORM=valu: the vector formulation of the fused pair, gauss_pair.inc -- exact, slower)
      static const bool valuForm = svdev::env("SSRLCV_GAUSS_PAIR_FORM") != nullptr && svdev::env("SSRLCV_GAUSS_PAIR_FORM")[0] == 'v';
      const float* pin = f32Pair ? in : nullptr;
      const uint8_t* pu8 = f32Pair ? nullptr : pixels;
      rc = valuForm ? launch_pair(pin, pu8, (float*)(ws + offGauss[0]), (float*)(ws + offGauss[1]), oc.w, oc.h, oc.weights[0], oc.weights[1], mm, mm + 2, so)
                    : launch_pair_rm(pin, pu8, (float*)(ws + offGauss[0]), (float*)(ws + offGauss[1]), oc.w, oc.h, oc.weights[0], oc.weights[1], mm, mm + 2, so);
      if (rc) return rc;
      src = (const float*)(ws + offGauss[1]);
      firstLevel = 2;
    }
    for (int b = firstLevel; b < svp::kGauss; ++b) {
      float* dst = (float*)(ws + offGauss[b]);
      // next octave input = 2x2 bin of the UN-normalised level 3 (src/FeatureFactory.cu:392-399): written by level 3's
      // convolution itself where that kernel can (see launch_conv), by k_bin2x otherwise
      bool binned = false;
      rc = launch_conv(src, dst, nullptr, oc.w, oc.h, oc.taps[b], oc.weights[b], mm + 2 * b, so,
                       (fuseUpsample && o == 0 && b == 0) ? pixels : nullptr,
                       (b == 3 && o + 1 < svp::kOctaves) ? nextIn[o] : nullptr, &binned);
      if (rc) return rc;
      src = dst;
      if (b == 3 && o + 1 < svp::kOctaves) {
        if (!binned) {
          rc = ssrlcv_hip_bin2x(lv[3], oc.w, oc.h, nextIn[o], (ssrlcv_stream_t)so);
          if (rc) return rc;
        }
        in = nextIn[o];
        if (as) SSRLCV_HIP_TRY(hipEventRecord(as->binDone[o], so));
        if (deferDog0 && as && o == deferDog0) {
          // Developer build, SSRLCV_DOGX0_AFTER=n: octave 0's DoG pass (no one in this stage waits for it) is held back until
          // level 3 of octave n is through, so that the octaves that do have successors get the machine first.  Exact, measured
          // and NOT the default: build_dog 1.70 -> 1.76 ms per 4096^2 image for n = 1 and 2 (profiles/r05_kernel_ab.txt)
          const svp::OctavePlan& o0 = plan->oct[0];
          const float* lv0[svp::kGauss];
          for (int q = 0; q < svp::kGauss; ++q) lv0[q] = (const float*)(ws + plan->off_gauss[0][q]);
          SSRLCV_HIP_TRY(hipStreamWaitEvent(sd, as->binDone[o], 0));
          rc = launch_dogx(lv0, mmAll, o0.w, o0.h, (uint8_t*)(ws + o0.off_flags), mmAll + 2 * svp::kGauss, (float*)(ws + plan->off_dogPartial), 0,
                           svp::kDog, 0, false, firstNoise, sched.waves, sd);
          if (rc) return rc;
          SSRLCV_HIP_TRY(hipEventRecord(as->dogDone[0], sd));
        }
      }
      if (split && b == 3) {  // DoG 0..2: extrema of level 1, min / max of 0..2
        SSRLCV_HIP_TRY(hipEventRecord(as->levelDone[o][b], so));
        SSRLCV_HIP_TRY(hipStreamWaitEvent(sdo, as->levelDone[o][b], 0));
        rc = launch_dogx(lv, mm, oc.w, oc.h, flags, mm + 2 * svp::kGauss, dogPartial, 0, 3, 0, false, firstNoise, sched.waves, sdo);
        if (rc) return rc;
      }
    }
    if (as) {
      SSRLCV_HIP_TRY(hipEventRecord(as->convDone[o], so));
      SSRLCV_HIP_TRY(hipStreamWaitEvent(sdo, as->convDone[o], 0));
    }
    if (deferDog0 && as && o == 0) continue;  // (its DoG pass is queued from octave deferDog0's iteration, see above)
    if (split) rc = launch_dogx(lv, mm, oc.w, oc.h, flags, mm + 2 * svp::kGauss, dogPartial, 1, svp::kDog, 3, true, firstNoise, sched.waves, sdo);
    else rc = launch_dogx(lv, mm, oc.w, oc.h, flags, mm + 2 * svp::kGauss, dogPartial, 0, svp::kDog, 0, false, firstNoise, sched.waves, sdo);
    if (rc) return rc;
    if (as) SSRLCV_HIP_TRY(hipEventRecord(as->dogDone[o], sdo));
#if SSRLCV_EARLY_CHAIN0
    // Fused extract (round 5): octave 0's list chain (60 % of the key points: ~0.3 ms of small launches) starts behind its DoG
    // pass on the one side stream build_dog does not use, beside the latency-bound tail of the small octaves, instead of
    // beside the HBM-bound gradient tables at the head of the key-point stage.  Exact (tests/test_gpu_sift.py); bench step
    // 9.99 / 10.00 -> 9.86 / 9.92 ms: the scale-space stage as the stage event sees it grows by 0.07 ms per 4096^2 image (its
    // tail shares the machine now), the key-point stage shrinks by 0.13.  Octave 1's chain as well (mask 3: 10.19 / 10.25)
    // or all four (mask 15: 10.27 / 10.26) lose: behind octave 0's chain on the same stream they give up their own.
    // Developer build: SSRLCV_NO_EARLY_CHAIN=1 keeps every chain in describe.
    static const bool noEarlyChain = svdev::env("SSRLCV_NO_EARLY_CHAIN") != nullptr;
    if (as && !noEarlyChain && ((SSRLCV_EARLY_CHAIN0 >> o) & 1) && plan->fusedCall && plan->stopStage >= 6) {
      SSRLCV_HIP_TRY(hipStreamWaitEvent(as->chain2, as->dogDone[o], 0));
      rc = svp::launch_chain_octave(plan, ws, o, as->chain2);
      if (rc) return rc;
      plan->chain0InFlight |= 1 << o;
    }
#endif
    if (earlyPolar) {  // fused extract: this octave's gradient tables start now, on their own stream
      SSRLCV_HIP_TRY(hipStreamWaitEvent(as->polar, as->dogDone[o], 0));
      svp::launch_polar_octave(plan, ws, o, as->polar);
      SSRLCV_HIP_TRY(hipEventRecord(as->polarDone[o], as->polar));
    }
  }
  if (earlyPolar) plan->polarInFlight = 1;
  if (as) {  // join: every stream a DoG pass ran on (in-order streams: the last pass of each covers the earlier ones)
    for (int o = 0; o < svp::kOctaves; ++o) SSRLCV_HIP_TRY(hipStreamWaitEvent(st, as->dogDone[o], 0));
  }
  return SSRLCV_OK;
}

}  // extern "C"
