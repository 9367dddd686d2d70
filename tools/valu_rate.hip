// Developer micro-benchmark: sustained v_fma_f32 issue rate per SIMD at a given occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int PK>
__global__ void k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
      x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
int main() {
  float* d;
  hipMalloc(&d, 256 * 4096 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wavesPerSimd = 1; wavesPerSimd <= 8; wavesPerSimd *= 2) {
    int blocks = 256 * wavesPerSimd;  // 256 threads = 4 waves = 1 per SIMD per block
    int iters = 20000;
    hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = (double)iters * 128 * wavesPerSimd;  // wave-instructions issued on each SIMD
    double tf = (double)blocks * 256 * iters * 128 * 2 / (ms * 1e-3) / 1e12;
    printf("waves/SIMD %d: %.3f ms  %.1f TFLOP/s  %.2f ns per wave-instr per SIMD (%.2f cycles at 2.4 GHz)\n", wavesPerSimd, ms, tf,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
  }
  return 0;
}
