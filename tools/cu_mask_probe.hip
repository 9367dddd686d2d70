// tools/cu_mask_probe.hip -- where do the workgroups of a CU-masked stream run?  (developer probe, MI355X)
// build: hipcc --offload-arch=gfx950 -O3 tools/cu_mask_probe.hip -o tools/_build/cu_mask_probe
// For a few masks of hipExtStreamCreateWithCUMask it launches 4096 one-wave workgroups that stamp the XCC_ID and HW_ID
// registers, and prints how many distinct (xcc, se, cu) places were used and the histogram per XCC -- the bit -> CU mapping
// is not documented for the eight-XCD part.  It then times a bandwidth-bound and an LDS-heavy kernel side by side on two
// complementary masks against the same pair on unmasked streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

#define CHECK(x)                                                                        \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); std::exit(1); } \
  } while (0)

__global__ void k_where(unsigned* out) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // a little work so that the blocks spread instead of all finishing on the first CU that frees up
  float a = (float)threadIdx.x;
  for (int i = 0; i < 2000; ++i) a = a * 1.0001f + 0.5f;
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = (xcc & 0xF) | (a == 0.123f ? 0x100u : 0u);
  }
}

static void place(const char* label, hipStream_t st, unsigned* dev, std::vector<unsigned>& host, int blocks) {
  hipLaunchKernelGGL(k_where, dim3(blocks), dim3(64), 0, st, dev);
  CHECK(hipStreamSynchronize(st));
  CHECK(hipMemcpy(host.data(), dev, sizeof(unsigned) * 2 * blocks, hipMemcpyDeviceToHost));
  std::set<unsigned> places;
  std::map<unsigned, std::set<unsigned>> perXcc;
  for (int b = 0; b < blocks; ++b) {
    const unsigned hw = host[2 * b], xcc = host[2 * b + 1] & 0xF;
    const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    const unsigned key = (xcc << 16) | (se << 8) | (sh << 4) | cu;
    places.insert(key);
    perXcc[xcc].insert(key);
  }
  std::printf("%-28s %3zu distinct CUs;", label, places.size());
  for (auto& kv : perXcc) std::printf(" xcc%u:%zu", kv.first, kv.second.size());
  std::printf("\n");
}

__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
// one block per CU's LDS (128 KB), FMA work
__global__ __launch_bounds__(512) void k_ldsheavy(float* out, int iters) {
  extern __shared__ float s[];
  for (int i = threadIdx.x; i < 32768; i += 512) s[i] = (float)i;
  __syncthreads();
  float a = 0.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll 8
    for (int k = 0; k < 64; ++k) a = __builtin_fmaf(s[(threadIdx.x + 64 * k + it) & 32767], 1.0001f, a);
  }
  if (a == 1.2345f) out[blockIdx.x] = a;
}

int main() {
  int dev = 0, cus = 0;
  CHECK(hipGetDevice(&dev));
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  std::printf("multiprocessors: %d\n", cus);
  const int blocks = 8192;
  unsigned* d = nullptr;
  CHECK(hipMalloc(&d, sizeof(unsigned) * 2 * blocks));
  std::vector<unsigned> h(2 * blocks);
  hipStream_t plain;
  CHECK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
  place("no mask", plain, d, h, blocks);
  struct M { const char* name; unsigned w[8]; };
  const M masks[] = {
      {"bits 0-63", {0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0, 0, 0}},
      {"bits 0-31", {0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0}},
      {"bits 0-7", {0xFFu, 0, 0, 0, 0, 0, 0, 0}},
      {"bits 64-255", {0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}},
      {"every 4th bit", {0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u}},
      {"low byte of each word", {0xFFu, 0xFFu, 0xFFu, 0xFFu, 0xFFu, 0xFFu, 0xFFu, 0xFFu}},
  };
  hipStream_t small = nullptr, large = nullptr;
  for (const M& m : masks) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, m.w);
    if (e != hipSuccess) { std::printf("%-28s hipExtStreamCreateWithCUMask -> %s\n", m.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
    place(m.name, st, d, h, blocks);
    if (m.w[0] == 0xFFFFFFFFu && m.w[1] == 0xFFFFFFFFu) small = st;
    else if (m.w[0] == 0 && m.w[2] == 0xFFFFFFFFu) large = st;
    else CHECK(hipStreamDestroy(st));
  }
  // side-by-side timing
  const size_t n = (size_t)64 << 20;  // float4: 1 GiB each way
  float4 *a = nullptr, *b = nullptr;
  float* o = nullptr;
  CHECK(hipMalloc(&a, n * 16));
  CHECK(hipMalloc(&b, n * 16));
  CHECK(hipMalloc(&o, 4096 * 4));
  CHECK(hipMemset(a, 0, n * 16));
  CHECK(hipFuncSetAttribute((const void*)k_ldsheavy, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  hipStream_t plain2;
  CHECK(hipStreamCreateWithFlags(&plain2, hipStreamNonBlocking));
  hipEvent_t e0, e1, f0, f1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&f0)); CHECK(hipEventCreate(&f1));
  auto pair = [&](const char* label, hipStream_t sa, hipStream_t sb, int heavyBlocks, int copyBlocks) {
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0, sa));
      hipLaunchKernelGGL(k_ldsheavy, dim3(heavyBlocks), dim3(512), 131072, sa, o, 600);
      CHECK(hipEventRecord(e1, sa));
      CHECK(hipEventRecord(f0, sb));
      hipLaunchKernelGGL(k_stream, dim3(copyBlocks), dim3(256), 0, sb, a, b, n);
      CHECK(hipEventRecord(f1, sb));
      CHECK(hipDeviceSynchronize());
    }
    float th = 0, tc = 0;
    CHECK(hipEventElapsedTime(&th, e0, e1));
    CHECK(hipEventElapsedTime(&tc, f0, f1));
    std::printf("%-44s lds-heavy %.3f ms, copy %.3f ms (%.0f GB/s)\n", label, th, tc, 2.0 * n * 16 / tc * 1e-6);
  };
  pair("alone: heavy 256 blocks | copy 2048 blocks", plain, plain, 256, 2048);
  pair("two plain streams, heavy 256 | copy 2048", plain, plain2, 256, 2048);
  if (small && large) {
    pair("heavy on bits 64-255 (192) | copy on 0-63", large, small, 192, 512);
    pair("heavy on bits 64-255 (256 blocks) | copy", large, small, 256, 512);
  }
  std::printf("done\n");
  return 0;
}
