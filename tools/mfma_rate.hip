// Developer micro-benchmark: issue cost of f32 / f16 MFMAs per SIMD with 1, 2, 4 independent accumulator chains and
// 1 / 2 waves per SIMD.  build: hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/_build/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int CH>
__global__ void k_f32_16x16x4(float* out, int iters, float a, float b) {
  f32x4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = f32x4{a, b, a, b};
  float x = a + threadIdx.x, y = b;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[c], 0, 0, 0);
    }
  }
  float s = 0;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH>
__global__ void k_f16_32x32x16(float* out, int iters, float a, float b) {
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c)
    for (int j = 0; j < 16; ++j) acc[c][j] = a;
  half8 x, y;
  for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(a + j); y[j] = (_Float16)b; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[c], 0, 0, 0);
    }
  }
  float s = 0;
  for (int c = 0; c < CH; ++c)
    for (int j = 0; j < 16; ++j) s += acc[c][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ void k_i8_32x32x32(float* out, int iters, float a, float b) {
  i32x16 acc[CH];
  for (int c = 0; c < CH; ++c)
    for (int j = 0; j < 16; ++j) acc[c][j] = (int)a;
  i32x4 x = {(int)threadIdx.x, 0x01020304, (int)b, 0x7f80ff01}, y = {0x01010101, (int)threadIdx.x, 0x02020202, 3};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, acc[c], 0, 0, 0);
    }
  }
  int s = 0;
  for (int c = 0; c < CH; ++c)
    for (int j = 0; j < 16; ++j) s += acc[c][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}
typedef void (*kern_t)(float*, int, float, float);
int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 4096 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  struct E { const char* name; kern_t fn; int ch; } tab[] = {
      {"v_mfma_f32_16x16x4_f32  1 chain ", k_f32_16x16x4<1>, 1}, {"v_mfma_f32_16x16x4_f32  2 chains", k_f32_16x16x4<2>, 2},
      {"v_mfma_f32_16x16x4_f32  4 chains", k_f32_16x16x4<4>, 4}, {"v_mfma_f32_32x32x16_f16 1 chain ", k_f16_32x32x16<1>, 1},
      {"v_mfma_f32_32x32x16_f16 2 chains", k_f16_32x32x16<2>, 2}, {"v_mfma_f32_32x32x16_f16 4 chains", k_f16_32x32x16<4>, 4},
      {"v_mfma_i32_32x32x32_i8  1 chain ", k_i8_32x32x32<1>, 1}, {"v_mfma_i32_32x32x32_i8  2 chains", k_i8_32x32x32<2>, 2},
      {"v_mfma_i32_32x32x32_i8  4 chains", k_i8_32x32x32<4>, 4}};
  for (const E& e : tab) {
    printf("%s", e.name);
    for (int w = 1; w <= 4; w *= 2) {
      int blocks = 256 * w, iters = 2000;
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0f, 0.5f);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, 0.5f);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      double per_simd = (double)iters * 16 * e.ch * w;
      printf("  w%d: %6.2f cyc/MFMA", w, ms * 1e6 / per_simd * 2.4);
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
