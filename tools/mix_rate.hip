// Developer micro-benchmark: does the f32 matrix pipe overlap with VALU work of the same SIMD?  Each iteration issues
// NV VALU instructions (two dependent chains of v_fma_f32) and NM MFMAs (v_mfma_f32_16x16x4_f32 or v_mfma_f32_4x4x1_16b_f32)
// on two alternating accumulators; prints cycles (at 2.4 GHz) per iteration per SIMD for 2 / 4 / 8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/mix_rate.hip -o tools/_build/mix_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NM, int KIND>
__global__ __launch_bounds__(256) void k_mix(float* out, int iters, float a, float b) {
  f32x4 acc0 = {a, b, a, b}, acc1 = {b, a, b, a};
  float x = a + threadIdx.x, y = b + threadIdx.x, p = a, q = b;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (v & 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(p), "v"(q));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y) : "v"(q), "v"(p));
      }
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        if (KIND == 0) {
          if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc1, 0, 0, 0);
          else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc0, 0, 0, 0);
        } else {
          if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, acc1, 0, 0, 0);
          else acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, acc0, 0, 0, 0);
        }
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3] + x + y;
}
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
// the int8 matrix pipe (v_mfma_i32_32x32x32_i8, 16 passes) beside NV VALU instructions
template <int NV>
__global__ __launch_bounds__(256) void k_mix_i8(float* out, int iters, float a, float b) {
  i32x16 acc0, acc1;
  for (int j = 0; j < 16; ++j) { acc0[j] = (int)a; acc1[j] = (int)b; }
  i32x4 xa = {(int)threadIdx.x, 0x01020304, (int)b, 0x7f80ff01}, xb = {0x01010101, (int)threadIdx.x, 0x02020202, 3};
  float x = a + threadIdx.x, y = b + threadIdx.x, p = a, q = b;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (v & 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(p), "v"(q));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y) : "v"(q), "v"(p));
      }
      if (u & 1) acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa, xb, acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa, xb, acc0, 0, 0, 0);
    }
  }
  int sum = 0;
  for (int j = 0; j < 16; ++j) sum += acc0[j] + acc1[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)sum + x + y;
}
typedef void (*kern_t)(float*, int, float, float);
int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 8192 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  struct E { const char* name; kern_t fn; } tab[] = {
      {"VALU 8            ", k_mix<8, 0, 0>},       {"16x16x4 x1        ", k_mix<0, 1, 0>},
      {"16x16x4 x1 + VALU4", k_mix<4, 1, 0>},       {"16x16x4 x1 + VALU8", k_mix<8, 1, 0>},
      {"16x16x4 x1 + VALU12", k_mix<12, 1, 0>},     {"4x4x1 x1          ", k_mix<0, 1, 1>},
      {"4x4x1 x2          ", k_mix<0, 2, 1>},       {"4x4x1 x2 + VALU4  ", k_mix<4, 2, 1>},
      {"4x4x1 x2 + VALU8  ", k_mix<8, 2, 1>},       {"4x4x1 x1 + VALU4  ", k_mix<4, 1, 1>},
      {"4x4x1 x1 + VALU3  ", k_mix<3, 1, 1>},       {"4x4x1 x4 + VALU13 ", k_mix<13, 4, 1>},
      {"i8 32x32x32 x1    ", k_mix_i8<0>},          {"i8 32x32x32 + VALU4", k_mix_i8<4>},
      {"i8 32x32x32 + VALU8", k_mix_i8<8>},         {"i8 32x32x32 + VALU16", k_mix_i8<16>}};
  for (const E& e : tab) {
    printf("%s", e.name);
    for (int w = 2; w <= 8; w *= 2) {
      int blocks = 256 * w, iters = 2000;
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0f, 0.5f);
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, 0.5f);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      // per SIMD: w waves, each iters*4 inner iterations
      double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 4 * w);
      printf("  w=%d: %7.1f cyc/iter/wave-slot", w, cyc);
    }
    printf("\n");
  }
  return 0;
}
