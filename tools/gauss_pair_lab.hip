// Developer lab for the fused two-level Gaussian (ssrlcv_amd/csrc/gauss_pair.inc): bit-compares levels 0 and 1 (and their
// min / max) from ONE k_gauss_pair launch with the two separate launches of launch_conv, with a float input and with the
// u8 image whose 2x upsample is the input (octave 0), and times both.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Iinclude -Issrlcv_amd/csrc \
//        tools/gauss_pair_lab.hip ssrlcv_amd/csrc/capi_common.hip -o tools/_build/gauss_pair_lab
// usage: gauss_pair_lab [width=8192] [height=width] [rowsPerBlock=0]
#include "../ssrlcv_amd/csrc/pyramid.hip"
#include "lab_stubs.h"
#include <vector>

static size_t diffs(const std::vector<float>& a, const std::vector<float>& b, size_t W, const char* what) {
  size_t d = 0, first = 0;
  for (size_t i = 0; i < a.size(); ++i)
    if (memcmp(&a[i], &b[i], 4)) { if (!d) first = i; ++d; }
  if (d) printf("   %s: %zu diffs, first at x=%zu y=%zu: %g vs %g\n", what, d, first % W, first / W, a[first], b[first]);
  return d;
}

int main(int argc, char** argv) {
  const uint32_t W = argc > 1 ? (uint32_t)atoi(argv[1]) : 8192;
  const uint32_t H = argc > 2 ? (uint32_t)atoi(argv[2]) : W;
  const int rowsForced = argc > 3 ? atoi(argv[3]) : 0;
  const size_t n = (size_t)W * H;
  float *in, *a0, *a1, *b0, *b1, *mm;
  uint8_t* u8;
  hipMalloc(&in, n * 4); hipMalloc(&a0, n * 4); hipMalloc(&a1, n * 4); hipMalloc(&b0, n * 4); hipMalloc(&b1, n * 4);
  hipMalloc(&mm, 64); hipMalloc(&u8, n / 4);
  std::vector<float> h(n);
  std::vector<uint8_t> hu(n / 4);
  uint32_t s = 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) * (1.0f / 65536.0f); }
  for (size_t i = 0; i < n / 4; ++i) { s = s * 1664525u + 1013904223u; hu[i] = (uint8_t)(s >> 24); }
  hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(u8, hu.data(), n / 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float wA[129], wB[129];
  const int tA = ssrlcv_gauss_kernel_host(0.70710678f, 0.5f, wA), tB = ssrlcv_gauss_kernel_host(1.0f, 0.5f, wB);
  if (tA != 13 || tB != 17) { printf("taps %d %d\n", tA, tB); return 1; }
  if (!pair_usable(W, H, tA, tB, in, a0, a1)) { printf("pair kernel not usable at %ux%u\n", W, H); return 1; }
  (void)rowsForced;
  std::vector<float> ra(n), rb(n), pa(n), pb(n);
  int bad = 0;
  for (int ups = 0; ups < 2; ++ups) {
    const uint8_t* src = ups ? u8 : nullptr;
    if (ups && !upsample_fusable(W, H, tA)) { printf("u8 source not fusable at this size\n"); continue; }
    float mmRef[4], mmPair[4], msRef, msPair;
    const float init[4] = {FLT_MAX, -FLT_MAX, FLT_MAX, -FLT_MAX};
    // reference: two launches
    hipMemset(a0, 0xff, n * 4); hipMemset(a1, 0xff, n * 4);
    hipMemcpy(mm, init, 16, hipMemcpyHostToDevice);
    int rc = launch_conv(in, a0, nullptr, W, H, tA, wA, mm, nullptr, src);
    if (!rc) rc = launch_conv(a0, a1, nullptr, W, H, tB, wB, mm + 2, nullptr);
    if (rc || hipDeviceSynchronize() != hipSuccess) { printf("reference failed rc %d: %s\n", rc, hipGetErrorString(hipGetLastError())); return 1; }
    hipMemcpy(mmRef, mm, 16, hipMemcpyDeviceToHost);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) { launch_conv(in, a0, nullptr, W, H, tA, wA, nullptr, nullptr, src); launch_conv(a0, a1, nullptr, W, H, tB, wB, nullptr, nullptr); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&msRef, e0, e1); msRef /= 5;
    for (int form = 0; form < 2; ++form) {  // 0: vector formulation (k_gauss_pair), 1: matrix pipe (k_gauss_pair_rm)
    if (form == 1 && !pair_rm_usable(W, H)) { printf("matrix-pipe form not usable at this size\n"); continue; }
    auto launch = [&](float* mA, float* mB) { return form ? launch_pair_rm(in, src, b0, b1, W, H, wA, wB, mA, mB, nullptr) : launch_pair(in, src, b0, b1, W, H, wA, wB, mA, mB, nullptr); };
    // fused
    hipMemset(b0, 0xff, n * 4); hipMemset(b1, 0xff, n * 4);
    hipMemcpy(mm, init, 16, hipMemcpyHostToDevice);
    rc = launch(mm, mm + 2);
    if (rc || hipDeviceSynchronize() != hipSuccess) { printf("pair failed rc %d: %s\n", rc, hipGetErrorString(hipGetLastError())); return 1; }
    hipMemcpy(mmPair, mm, 16, hipMemcpyDeviceToHost);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch(nullptr, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&msPair, e0, e1); msPair /= 5;
    hipMemcpy(ra.data(), a0, n * 4, hipMemcpyDeviceToHost); hipMemcpy(rb.data(), a1, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(pa.data(), b0, n * 4, hipMemcpyDeviceToHost); hipMemcpy(pb.data(), b1, n * 4, hipMemcpyDeviceToHost);
#ifdef SSRLCV_STAMPS
    if (form == 1) {  // one more launch with s_memtime stamps in the middle block
      static long long* stamps = nullptr;
      const size_t ns = 8 * 64 * 8;
      if (!stamps) hipMalloc(&stamps, ns * 8);
      hipMemset(stamps, 0, ns * 8);
      g_lab_stamps = stamps;
      launch(nullptr, nullptr);
      hipDeviceSynchronize();
      g_lab_stamps = nullptr;
      std::vector<long long> st(ns);
      hipMemcpy(st.data(), stamps, ns * 8, hipMemcpyDeviceToHost);
      for (int wv = 0; wv < 4; wv += 3) {
        double d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int cnt = 0;
        for (int it = 6; it < 20; ++it) {
          const long long* s0 = &st[((size_t)wv * 64 + it) * 8];
          const long long* s1 = &st[((size_t)wv * 64 + it + 1) * 8];
          if (!s0[0] || !s1[0] || !s0[7]) continue;
          ++cnt;
          for (int k = 0; k < 7; ++k) d[k] += s0[k + 1] - s0[k];
          d[7] += s1[0] - s0[0];
        }
        if (cnt) printf("   wave %d: H_A %.0f | H_B %.0f | barrier %.0f | bv + stage + fetch %.0f | V_A %.0f | V_B %.0f | barrier %.0f | step %.0f clocks\n", wv, d[0] / cnt, d[1] / cnt, d[2] / cnt,
                        d[3] / cnt, d[4] / cnt, d[5] / cnt, d[6] / cnt, d[7] / cnt);
      }
    }
#endif
    const size_t dA = diffs(ra, pa, W, "level a"), dB = diffs(rb, pb, W, "level b");
    const bool mmOk = memcmp(mmRef, mmPair, 16) == 0;
    const double bytes = (ups ? n / 4.0 : n * 4.0) + 2.0 * n * 4;
    printf("%s input %ux%u %s: two launches %.3f ms, fused %.3f ms (%.2f TB/s on %.0f MB)  diffs a %zu b %zu  minmax %s (%g %g %g %g | %g %g %g %g)\n", ups ? "u8 " : "f32", W, H, form ? "matrix" : "vector", msRef, msPair,
           bytes / (msPair * 1e-3) / 1e12, bytes / 1e6, dA, dB, mmOk ? "equal" : "DIFFERENT", mmRef[0], mmRef[1], mmRef[2], mmRef[3], mmPair[0], mmPair[1], mmPair[2], mmPair[3]);
    bad += dA != 0 || dB != 0 || !mmOk;
    }
  }
  printf(bad ? "FAILED\n" : "all equal\n");
  return bad != 0;
}
