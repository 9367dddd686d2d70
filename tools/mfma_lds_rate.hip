// Developer micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 when every MFMA takes a fresh A operand from LDS
// (the banded-Toeplitz Gaussian's inner loop), against the same loop on register operands.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_lds_rate.hip -o tools/_build/mfma_lds_rate
// Prints cycles per MFMA per SIMD (2.4 GHz assumed) for 1 and 2 waves per SIMD; ideal = 32.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int KS = 20, TPW = 4;

// MODE 0: register operands; 1: LDS operands, vertical-pass pattern (row stride 272, 4 tiles 16 floats apart), read two
// k-steps ahead; 2: the same, read four k-steps ahead; 3: horizontal-pass pattern (row stride 354, k-steps 4 floats apart);
// 4: vertical pattern through five tile bases and 16-bit immediate offsets (single ds_read_b32, no address VALU in the loop)
template <int MODE>
__global__ __launch_bounds__(512) void k_loop(float* out, int iters, float a) {
  extern __shared__ float s_mem[];
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 36 * 1024; i += blockDim.x) s_mem[i] = a + i;
  __syncthreads();
  float tz[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) tz[s] = a * (s + 1) + lane;
  f32x4 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) acc[t] = f32x4{0, 0, 0, 0};
  const float* base = MODE == 3 ? s_mem + li * 354 + lk + (wave & 3) * 64 : s_mem + lk * 272 + li + (wave & 3) * 64;
  constexpr int AHEAD = MODE == 2 ? 4 : 2;
  for (int it = 0; it < iters; ++it) {
    const float* p = base + (it & 3) * 16;
    if (MODE == 0) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(tz[(ks + t) % KS], tz[ks], acc[t], 0, 0, 0);
    } else {
      float v[AHEAD + 1][TPW];
      const float* tb[5];
#pragma unroll
      for (int c = 0; c < 5; ++c) tb[c] = p + c * 16 * 272;
      const float* rowp[KS];
      if (MODE == 5) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) rowp[ks] = tb[(4 * ks) / 16] + ((4 * ks) % 16) * 272;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(rowp[ks]));
      }
      auto load = [&](int ks, float (&d)[TPW]) {
        if (MODE == 5) {  // per-k-step row pointers, all computed (and pinned) before the MFMA loop
          const float* q = rowp[ks];
#pragma unroll
          for (int t = 0; t < TPW; ++t) d[t] = q[t * 16];
          return;
        }
        if (MODE == 4) {  // per-tile bases + 16-bit immediates, volatile so that the reads stay single ds_read_b32
          const volatile float* q = tb[(4 * ks) / 16] + ((4 * ks) % 16) * 272;
#pragma unroll
          for (int t = 0; t < TPW; ++t) d[t] = q[t * 16];
          return;
        }
        const float* q = MODE == 3 ? p + 4 * ks : MODE == 6 ? p + ((4 * ks) % 16) * 272 : MODE == 7 ? p + 24576 + ((4 * ks) % 16) * 272 : p + (4 * ks) * 272;
#pragma unroll
        for (int t = 0; t < TPW; ++t) d[t] = q[t * 16];
      };
#pragma unroll
      for (int k = 0; k < AHEAD; ++k) load(k, v[k]);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + AHEAD < KS) load(ks + AHEAD, v[(ks + AHEAD) % (AHEAD + 1)]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[ks % (AHEAD + 1)][t], tz[ks], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int t = 0; t < TPW; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef void (*kern_t)(float*, int, float);
int main() {
  float* d;
  (void)hipMalloc(&d, 512 * 256 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  struct E { const char* name; kern_t fn; } tab[] = {{"register operands          ", k_loop<0>},
                                                     {"LDS, vertical pattern, +2  ", k_loop<1>},
                                                     {"LDS, vertical pattern, +4  ", k_loop<2>},
                                                     {"LDS, horizontal pattern, +2", k_loop<3>},
                                                     {"LDS, vertical, b32 + imm   ", k_loop<4>},
                                                     {"LDS, vertical, row pointers", k_loop<5>},
 {"LDS, vertical, 16-row span ", k_loop<6>}, {"LDS, vertical, 16 rows @96K", k_loop<7>}};
  for (const E& e : tab) {
    (void)hipFuncSetAttribute((const void*)e.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    printf("%s", e.name);
    for (int threads = 256; threads <= 512; threads *= 2) {
      const int iters = 400;
      hipLaunchKernelGGL(e.fn, dim3(256), dim3(threads), 150 * 1024, 0, d, 10, 1.0f);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(e.fn, dim3(256), dim3(threads), 150 * 1024, 0, d, iters, 1.0f);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double mfmaPerSimd = (double)iters * KS * TPW * (threads / 256);
      printf("  %d waves/SIMD: %.1f cycles/MFMA", threads / 256, ms * 1e-3 * 2.4e9 / mfmaPerSimd);
    }
    printf("\n");
  }
  return 0;
}
