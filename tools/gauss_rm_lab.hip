// Developer lab for the register-marching Gaussian (ssrlcv_amd/csrc/gauss_rm.inc): bit-compares it with the established
// kernels (output level, its min / max, the folded 2x2 bin) for the six radii of the sigma ladder and times both.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Iinclude -Issrlcv_amd/csrc \
//        tools/gauss_rm_lab.hip ssrlcv_amd/csrc/capi_common.hip -o tools/_build/gauss_rm_lab
// usage: gauss_rm_lab [width=8192] [height=width] [rowsPerBlock=0]
#include "../ssrlcv_amd/csrc/pyramid.hip"
#include "lab_stubs.h"
#include <vector>

int main(int argc, char** argv) {
  const uint32_t W = argc > 1 ? (uint32_t)atoi(argv[1]) : 8192;
  const uint32_t H = argc > 2 ? (uint32_t)atoi(argv[2]) : W;
  rm_rows() = argc > 3 ? atoi(argv[3]) : 0;
  rm_min_px() = 0;
  rm_oneb_mask() = getenv("RM_LAB_ONEB") ? atoi(getenv("RM_LAB_ONEB")) : 0;
  const size_t n = (size_t)W * H;
  float *in, *outA, *outB, *binA, *binB, *mm;
  hipMalloc(&in, n * 4); hipMalloc(&outA, n * 4); hipMalloc(&outB, n * 4);
  hipMalloc(&binA, n); hipMalloc(&binB, n); hipMalloc(&mm, 64);
  std::vector<float> h(n);
  uint32_t s = 12345u;
  const bool ramp = getenv("RM_LAB_RAMP") != nullptr;  // in = x + 1000 y: a shifted result shows its shift
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ramp ? (float)(i % W) + 1000.0f * (float)(i / W) : (float)(s >> 8) * (1.0f / 65536.0f); }
  hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const float sigmas[6] = {0.70710678f, 1.0f, 1.41421356f, 2.0f, 2.82842712f, 4.0f};
  std::vector<float> a(n), b(n), ba(n / 4), bb(n / 4);
#ifdef SSRLCV_STAMPS
  long long* stamps;
  const size_t ns = 8 * 64 * 8;
  hipMalloc(&stamps, ns * 8);
#endif
  int bad = 0;
  for (int lv = 0; lv < 6; ++lv) {
    float w[129];
    const int taps = ssrlcv_gauss_kernel_host(sigmas[lv], 0.5f, w);
    const int ksz = taps | 1;
    float mmA[2], mmB[2];
    bool binnedA = false, binnedB = false;
    float ms[2];
    for (int v = 0; v < 2; ++v) {
      rm_mask() = v ? 63 : 0;
      float* out = v ? outB : outA;
      float* bin = v ? binB : binA;
      hipMemset(out, 0xff, n * 4);
      hipMemset(bin, 0xff, n);
      const float init[2] = {FLT_MAX, -FLT_MAX};
      hipMemcpy(mm, init, 8, hipMemcpyHostToDevice);
      int rc = launch_conv(in, out, nullptr, W, H, ksz, w, mm, nullptr, nullptr, bin, v ? &binnedB : &binnedA);
      if (rc) { printf("launch_conv rc %d\n", rc); return 1; }
      if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
      hipMemcpy(v ? mmB : mmA, mm, 8, hipMemcpyDeviceToHost);
      hipEventRecord(e0);
      for (int r = 0; r < 5; ++r) launch_conv(in, out, nullptr, W, H, ksz, w, nullptr, nullptr, nullptr, bin, nullptr);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[v], e0, e1);
      ms[v] /= 5;
    }
#ifdef SSRLCV_STAMPS
    {  // one more launch of the register-marching kernel with s_memtime stamps (100 MHz ticks) in one block
      hipMemset(stamps, 0, ns * 8);
      g_lab_stamps = stamps;
      rm_mask() = 63;
      launch_conv(in, outB, nullptr, W, H, ksz, w, nullptr, nullptr, nullptr, binB, nullptr);
      hipDeviceSynchronize();
      g_lab_stamps = nullptr;
      std::vector<long long> st(ns);
      hipMemcpy(st.data(), stamps, ns * 8, hipMemcpyDeviceToHost);
      for (int wave = 0; wave < 4; wave += 3) {
        double d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int cnt = 0;
        for (int it = 6; it < 20; ++it) {
          const long long* s0 = &st[((size_t)wave * 64 + it) * 8];
          const long long* s1 = &st[((size_t)wave * 64 + it + 1) * 8];
          if (!s0[0] || !s1[0] || !s0[4]) continue;
          ++cnt;
          d[0] += s0[1] - s0[0]; d[1] += s0[2] - s0[1]; d[2] += s0[3] - s0[2]; d[3] += s0[4] - s0[3]; d[4] += s1[0] - s0[0];
          d[5] += s0[5] - s0[2]; d[6] += s0[6] - s0[5]; d[7] += s0[7] - s0[6];
        }
        if (cnt) printf("   wave %d: H %.0f | barrier A %.0f | V %.0f (stage write %.0f, fetch issue %.0f, first quantum %.0f) | barrier B %.0f | step %.0f clocks\n", wave,
                        d[0] / cnt, d[1] / cnt, d[2] / cnt, d[5] / cnt, d[6] / cnt, d[7] / cnt, d[3] / cnt, d[4] / cnt);
      }
    }
#endif
    hipMemcpy(a.data(), outA, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), outB, n * 4, hipMemcpyDeviceToHost);
    size_t diff = 0, first = 0;
    for (size_t i = 0; i < n; ++i)
      if (memcmp(&a[i], &b[i], 4)) { if (!diff) first = i; ++diff; }
    size_t bdiff = 0;
    if (binnedA && binnedB) {
      hipMemcpy(ba.data(), binA, n, hipMemcpyDeviceToHost);
      hipMemcpy(bb.data(), binB, n, hipMemcpyDeviceToHost);
      for (size_t i = 0; i < n / 4; ++i) bdiff += memcmp(&ba[i], &bb[i], 4) != 0;
    }
    const bool mmOk = memcmp(mmA, mmB, 8) == 0;
    if (ramp) {
      const size_t pts[6][2] = {{700, 300}, {701, 300}, {700, 301}, {1000, 517}, {1023, 600}, {1024, 600}};
      for (auto& pt : pts) printf("   (%zu, %zu): established %.3f  register-marching %.3f\n", pt[0], pt[1], a[pt[1] * W + pt[0]], b[pt[1] * W + pt[0]]);
    }
    printf("taps %2d  %ux%u: established %.3f ms, register-marching %.3f ms (%.2f TB/s of copy traffic)  diffs %zu (first at x=%zu y=%zu: %g vs %g)  bin %s diffs %zu  minmax %s\n",
           ksz, W, H, ms[0], ms[1], 2.0 * n * 4 / (ms[1] * 1e-3) / 1e12, diff, first % W, first / W, diff ? a[first] : 0.f, diff ? b[first] : 0.f,
           binnedA && binnedB ? "both" : "-", bdiff, mmOk ? "equal" : "DIFFERENT");
    bad += diff != 0 || bdiff != 0 || !mmOk;
  }
  printf(bad ? "FAILED\n" : "all equal\n");
  return bad != 0;
}
