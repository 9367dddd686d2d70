// Developer probe for the register-marching Gaussian (k_gauss_rm): (1) operand layout and rounding of
// v_mfma_f32_4x4x1_16b_f32 -- is D[i][j] of block b exactly fmaf(A[lane 4b+i], B[lane 4b+j], C) in lane 4b+j, register i,
// and is a chain of them a k-ordered fmaf chain; (2) the rate of the "one image row per lane" load shape (64 lanes x 16
// bytes from 64 different rows, marching along x) that its horizontal pass uses, with the strip / halo geometry of the kernel.
// build: hipcc --offload-arch=gfx950 -O3 tools/rm_probe.hip -o tools/_build/rm_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(const float* a, const float* b, const float* c, float* d, int chain) {
  const int l = threadIdx.x;
  f32x4 acc = {c[4 * l], c[4 * l + 1], c[4 * l + 2], c[4 * l + 3]};
  for (int k = 0; k < chain; ++k) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k * 64 + l], b[k * 64 + l], acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) d[4 * l + i] = acc[i];
}

// one row per lane: wave w of a 4-wave block owns output columns x0 + 64 w .. + 63 of a 256-column strip and reads
// columns x0 + 64 w - RP .. x0 + 64 w + 63 + RP of the 64 rows of a step, 16 bytes per lane per load, BATCH loads (a run of
// 16 BATCH bytes per row) issued together
template <int RP, int BATCH>
__global__ __launch_bounds__(256) void k_rowload(const float* __restrict__ in, int W, int H, int rowsPerBlock, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x0 = blockIdx.x * 256 + 64 * wave - RP;
  const int y0 = blockIdx.y * rowsPerBlock;
  constexpr int NL = (64 + 2 * RP) / 4;  // loads per lane per step
  static_assert(NL % BATCH == 0, "");
  f32x4 acc = {0, 0, 0, 0};
  for (int ys = y0; ys < y0 + rowsPerBlock; ys += 64) {
    const int y = ys + lane < H ? ys + lane : H - 1;
    const float* row = in + (size_t)y * W;
    f32x4 v[2][BATCH];
    auto fetch = [&](int bi, f32x4 (&dst)[BATCH]) {
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        int x = x0 + 4 * (bi * BATCH + u);
        x = x < 0 ? 0 : (x > W - 4 ? W - 4 : x);
        dst[u] = *reinterpret_cast<const f32x4*>(row + x);
      }
    };
    fetch(0, v[0]);
#pragma unroll
    for (int bi = 0; bi < NL / BATCH; ++bi) {
      if (bi + 1 < NL / BATCH) fetch(bi + 1, v[(bi + 1) & 1]);
#pragma unroll
      for (int u = 0; u < BATCH; ++u) acc += v[bi & 1][u];
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.0f;
}

// the same bytes, coalesced (one row segment per wave instruction): the reference point
template <int RP>
__global__ __launch_bounds__(256) void k_rowload_coalesced(const float* __restrict__ in, int W, int H, int rowsPerBlock, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x0 = blockIdx.x * 256;
  const int y0 = blockIdx.y * rowsPerBlock;
  f32x4 acc = {0, 0, 0, 0};
  for (int ys = y0; ys < y0 + rowsPerBlock; ys += 64) {
#pragma unroll 4
    for (int r = wave; r < 64; r += 4) {
      const int y = ys + r < H ? ys + r : H - 1;
      acc += *reinterpret_cast<const f32x4*>(in + (size_t)y * W + x0 + 4 * lane);
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.0f;
}

int main() {
  // ---- (1) layout / rounding
  {
    const int chain = 68;
    std::vector<float> a(64 * chain), b(64 * chain), c(256), d(256);
    srand(7);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.0f - 1.0f; };
    for (auto& v : a) v = rnd();
    for (auto& v : b) v = rnd() * 255.0f;
    for (auto& v : c) v = rnd();
    float *da, *db, *dc, *dd;
    (void)hipMalloc(&da, a.size() * 4); (void)hipMalloc(&db, b.size() * 4); (void)hipMalloc(&dc, 1024); (void)hipMalloc(&dd, 1024);
    (void)hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dc, c.data(), 1024, hipMemcpyHostToDevice);
    for (int n : {1, chain}) {
      hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, da, db, dc, dd, n);
      (void)hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
      int badAB = 0, badBA = 0;
      for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
          float e1 = c[4 * l + i], e2 = c[4 * l + i];
          for (int k = 0; k < n; ++k) {
            e1 = fmaf(a[k * 64 + (l & ~3) + i], b[k * 64 + l], e1);  // D[i][j]: A row i, B column j = lane
            e2 = fmaf(a[k * 64 + l], b[k * 64 + (l & ~3) + i], e2);  // the transposed reading
          }
          badAB += memcmp(&e1, &d[4 * l + i], 4) != 0;
          badBA += memcmp(&e2, &d[4 * l + i], 4) != 0;
        }
      printf("4x4x1 chain of %2d: mismatches with D[reg i][lane j] = fma(A[lane 4b+i], B[lane j]): %d   transposed reading: %d\n", n, badAB, badBA);
    }
  }
  // ---- (2) row-per-lane load rate
  const int W = 8192, H = 8192;
  float *img, *out;
  (void)hipMalloc(&img, (size_t)W * H * 4);
  (void)hipMalloc(&out, 64);
  (void)hipMemset(img, 0, (size_t)W * H * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto launch) {
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("%-44s %.3f ms   %.2f TB/s of unique bytes\n", name, ms, (double)W * H * 4 / (ms * 1e-3) / 1e12);
  };
  for (int rpb : {512, 1024}) {
    dim3 grid(W / 256, H / rpb);
    printf("rows per block %d (%d blocks)\n", rpb, grid.x * grid.y);
    timeit("coalesced rows, no halo", [&] { hipLaunchKernelGGL(k_rowload_coalesced<16>, grid, dim3(256), 0, 0, img, W, H, rpb, out); });
    timeit("row per lane  RP  8 batch 4", [&] { hipLaunchKernelGGL((k_rowload<8, 4>), grid, dim3(256), 0, 0, img, W, H, rpb, out); });
    timeit("row per lane  RP 16 batch 4", [&] { hipLaunchKernelGGL((k_rowload<16, 4>), grid, dim3(256), 0, 0, img, W, H, rpb, out); });
    timeit("row per lane  RP 16 batch 8", [&] { hipLaunchKernelGGL((k_rowload<16, 8>), grid, dim3(256), 0, 0, img, W, H, rpb, out); });
    timeit("row per lane  RP 16 batch 2", [&] { hipLaunchKernelGGL((k_rowload<16, 2>), grid, dim3(256), 0, 0, img, W, H, rpb, out); });
    timeit("row per lane  RP 32 batch 8", [&] { hipLaunchKernelGGL((k_rowload<32, 8>), grid, dim3(256), 0, 0, img, W, H, rpb, out); });
  }
  return 0;
}
