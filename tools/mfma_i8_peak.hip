// tools/mfma_i8_peak.hip -- what the int8 matrix pipe sustains on this chip when nothing else is in the way: every wave
// issues independent v_mfma_i32_32x32x32_i8 back to back from registers (no LDS, no memory, no vector instructions in the
// loop), two waves per SIMD, for a few milliseconds; operands are zeros or random bytes (the clock the part holds under
// this load depends on how many bits toggle).  The matcher's roofline is priced against the 5 POP/s of the data sheet
// (2.4 GHz); this is the ceiling the same instruction reaches in practice.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_i8_peak.hip -o tools/_build/mfma_i8_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256, 2) void k_peak(const i32x4* __restrict__ src, int iters, int* __restrict__ sink) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  i32x4 a[4], b[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) { a[s] = src[(t * 8 + s) & 65535]; b[s] = src[(t * 8 + 4 + s) & 65535]; }
  i32x16 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], b[(s + c) & 3], acc[c], 0, 0, 0);
  }
  int v = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) v ^= acc[c][r];
  if (v == 0x12345678) sink[0] = v;
}

int main() {
  const size_t words = 65536;
  std::vector<int> host(words * 4);
  i32x4* src = nullptr;
  int* sink = nullptr;
  hipMalloc(&src, words * 16);
  hipMalloc(&sink, 4);
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int blocks = cus * 2;  // two blocks of four waves per CU: two waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int pattern = 0; pattern < 2; ++pattern) {
    srand(1);
    for (size_t i = 0; i < host.size(); ++i) host[i] = pattern ? (int)((unsigned)rand() * 2654435761u) : 0;
    hipMemcpy(src, host.data(), words * 16, hipMemcpyHostToDevice);
    for (int chains = 1; chains <= 4; chains *= 2) {
      const int iters = 40000 / chains;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (chains == 1) hipLaunchKernelGGL(k_peak<1>, dim3(blocks), dim3(256), 0, 0, src, iters, sink);
        else if (chains == 2) hipLaunchKernelGGL(k_peak<2>, dim3(blocks), dim3(256), 0, 0, src, iters, sink);
        else hipLaunchKernelGGL(k_peak<4>, dim3(blocks), dim3(256), 0, 0, src, iters, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const double ops = (double)blocks * 4 * iters * 4 * chains * 2.0 * 32 * 32 * 32;
      std::printf("%-7s operands, %d chain(s) per wave, 2 waves per SIMD: %.2f ms, %.2f POP/s\n", pattern ? "random" : "zero", chains, ms,
                  ops / (ms * 1e-3) / 1e15);
    }
  }
  return 0;
}
