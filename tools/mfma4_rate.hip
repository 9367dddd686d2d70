// Developer micro-benchmark: issue rate of v_mfma_f32_4x4x1_16b_f32 per SIMD (independent accumulators, operands in
// registers) at 1 / 2 / 4 waves per SIMD, and the rate of the s_memtime counter against the wall clock.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma4_rate.hip -o tools/_build/mfma4_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ __launch_bounds__(256) void k_rate(float* out, long long* ticks, int iters, float a, float b) {
  f32x4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = f32x4{a, b, a, b};
  float x = a + threadIdx.x, y = b + threadIdx.x;
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, acc[c], 0, 0, 0);
    }
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  float s = 0;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

int main() {
  float* d;
  long long* t;
  (void)hipMalloc(&d, 256 * 4096 * 4);
  (void)hipMalloc(&t, 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 4000, CH = 8;
  for (int w = 1; w <= 4; w *= 2) {
    const int blocks = 256 * w;  // w blocks of 4 waves per CU = w waves per SIMD
    hipLaunchKernelGGL(k_rate<CH>, dim3(blocks), dim3(256), 0, 0, d, t, iters, 1.0f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<CH>, dim3(blocks), dim3(256), 0, 0, d, t, iters, 1.0f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long ticks;
    (void)hipMemcpy(&ticks, t, 8, hipMemcpyDeviceToHost);
    const double mfmaPerSimd = (double)iters * 8 * CH * w;
    printf("%d wave(s) per SIMD: %.3f ms, %.2f ns per 4x4x1 MFMA per SIMD (%.1f TFLOP/s on 1024 SIMDs), s_memtime %.0f ticks per us, %.2f ticks per MFMA of one wave\n",
           w, ms, ms * 1e6 / mfmaPerSimd, 512.0 / (ms * 1e6 / mfmaPerSimd) * 1024 / 1e3, ticks / (ms * 1e3), (double)ticks / (iters * 8.0 * CH));
  }
  return 0;
}
