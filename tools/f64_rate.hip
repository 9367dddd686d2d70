// Developer micro-benchmark: issue cost of the f64 instructions sv_math.h compiles to (per SIMD, cycles at 2.4 GHz, at
// 1/2/4/8 waves per SIMD), plus an exhaustive check of v_cvt_rpi_i32_f32 against exact round-half-up for every
// non-negative float below 2^31.
// build: hipcc --offload-arch=gfx950 -O3 tools/f64_rate.hip -o tools/_build/f64_rate
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>

#define CHAIN8(OP) OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)
#define DEF_KERNEL(NAME, OP)                                                                                            \
  __global__ void NAME(double* out, int iters, double a, double b) {                                                    \
    double x0 = threadIdx.x + 1.0, x1 = threadIdx.x + 2.0, x2 = threadIdx.x + 3.0, x3 = threadIdx.x + 4.0,              \
           x4 = threadIdx.x + 5.0, x5 = threadIdx.x + 6.0, x6 = threadIdx.x + 7.0, x7 = threadIdx.x + 8.0;              \
    float f = (float)threadIdx.x;                                                                                      \
    int ii = iters;                                                                                                    \
    for (int i = 0; i < iters; ++i) {                                                                                  \
      _Pragma("unroll") for (int u = 0; u < 16; ++u) { CHAIN8(OP) }                                                    \
    }                                                                                                                  \
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + f + ii;                        \
  }
#define OP_FMA(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define OP_MUL(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(a));
#define OP_ADD(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_RCP(x) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
#define OP_RNDNE(x) asm volatile("v_rndne_f64 %0, %0" : "+v"(x));
#define OP_LDEXP(x) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x) : "v"(ii));
#define OP_CVT_F64_F32(x) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(x) : "v"(f));
#define OP_CVT_F32_F64(x) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(f) : "v"(x));
#define OP_CVT_I32_F64(x) asm volatile("v_cvt_i32_f64 %0, %1" : "+v"(ii) : "v"(x));
#define OP_CMP(x) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
#define OP_CNDMASK2(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f) : "v"(f));
#define OP_FMA32(x) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f) : "v"(f));
DEF_KERNEL(k_fma, OP_FMA)
DEF_KERNEL(k_mul, OP_MUL)
DEF_KERNEL(k_add, OP_ADD)
DEF_KERNEL(k_rcp, OP_RCP)
DEF_KERNEL(k_rndne, OP_RNDNE)
DEF_KERNEL(k_ldexp, OP_LDEXP)
DEF_KERNEL(k_cvt_f64_f32, OP_CVT_F64_F32)
DEF_KERNEL(k_cvt_f32_f64, OP_CVT_F32_F64)
DEF_KERNEL(k_cvt_i32_f64, OP_CVT_I32_F64)
DEF_KERNEL(k_cmp, OP_CMP)
DEF_KERNEL(k_fma32, OP_FMA32)

__global__ void k_rpi_check(unsigned long long* bad, unsigned* firstBad) {
  const unsigned long long n = 0x4F000000ull;  // bit patterns of the floats in [0, 2^31)
  unsigned long long local = 0;
  for (unsigned long long p = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; p < n;
       p += (unsigned long long)gridDim.x * blockDim.x) {
    const float x = __builtin_bit_cast(float, (unsigned)p);
    int r;
    asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    const float fl = floorf(x);
    const long long want = (long long)fl + ((x - fl) >= 0.5f ? 1 : 0);
    if ((long long)r != want) {
      ++local;
      atomicMin(firstBad, (unsigned)p);
    }
  }
  // the window coordinates of the descriptor kernel can dip to (-1, 0): same check there
  for (unsigned long long p = 0x80000000ull + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; p <= 0xBF800000ull;
       p += (unsigned long long)gridDim.x * blockDim.x) {
    const float x = __builtin_bit_cast(float, (unsigned)p);
    int r;
    asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    const float fl = floorf(x);
    const long long want = (long long)fl + ((x - fl) >= 0.5f ? 1 : 0);
    if ((long long)r != want) {
      ++local;
      atomicMin(firstBad, (unsigned)p);
    }
  }
  if (local) atomicAdd(bad, local);
}

template <typename K>
static void run(const char* name, K kern, double* out) {
  const int iters = 2000;
  for (int wps = 1; wps <= 8; wps *= 2) {
    dim3 grid(256 * 4), block(64 * wps);  // one block per SIMD-quad: 256 CUs x 4 blocks x wps waves... coarse but uniform
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, 10, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, iters, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: (1024 blocks x wps waves / 1024 SIMDs) x iters x 128
    const double perSimd = (double)wps * iters * 128.0;
    const double ns = ms * 1e6 / perSimd;
    printf("%-16s %d waves/SIMD: %.3f ns = %.2f cycles per wave-instruction per SIMD\n", name, wps, ns, ns * 2.4);
  }
}

int main() {
  double* out;
  hipMalloc(&out, sizeof(double) * 1024 * 512);
  run("v_fma_f64", k_fma, out);
  run("v_mul_f64", k_mul, out);
  run("v_add_f64", k_add, out);
  run("v_rcp_f64", k_rcp, out);
  run("v_rndne_f64", k_rndne, out);
  run("v_ldexp_f64", k_ldexp, out);
  run("v_cvt_f64_f32", k_cvt_f64_f32, out);
  run("v_cvt_f32_f64", k_cvt_f32_f64, out);
  run("v_cvt_i32_f64", k_cvt_i32_f64, out);
  run("v_cmp_gt_f64", k_cmp, out);
  run("v_fma_f32", k_fma32, out);
  unsigned long long* bad;
  unsigned* firstBad;
  hipMalloc(&bad, 8);
  hipMalloc(&firstBad, 4);
  hipMemset(bad, 0, 8);
  hipMemset(firstBad, 0xFF, 4);
  hipLaunchKernelGGL(k_rpi_check, dim3(4096), dim3(256), 0, 0, bad, firstBad);
  unsigned long long hb;
  unsigned hf;
  hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
  hipMemcpy(&hf, firstBad, 4, hipMemcpyDeviceToHost);
  float ff;
  memcpy(&ff, &hf, 4);
  printf("v_cvt_rpi_i32_f32 vs exact round-half-up over [-1, 0] and [0, 2^31): %llu mismatches (first at bits 0x%08x = %.9g)\n", hb, hf,
         hb ? ff : 0.0f);
  return 0;
}
