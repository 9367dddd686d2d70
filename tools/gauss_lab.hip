// Developer lab: phase timing inside one block of the marching MFMA Gaussian kernel (s_memtime stamps per wave and step).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DSSRLCV_STAMPS -DSSRLCV_INSTRUMENTED_BUILD -Iinclude -Issrlcv_amd/csrc \
//        tools/gauss_lab.hip ssrlcv_amd/csrc/capi_common.hip -o tools/_build/gauss_lab
// usage: gauss_lab [size=8192] [stamps=0] ; prints, for taps 23/33/47/65, the kernel time and the average cycles between stamps
// horizontal-role waves: 0 top, 1 after the MFMA loop, 2 after the ring write, 3 after the stage write, 4 after the next fetch
// vertical-role waves:   0 top, 1 after the store of the previous result, 2 after the MFMA loop
#include "../ssrlcv_amd/csrc/pyramid.hip"
#include "lab_stubs.h"
#include <vector>

int main(int argc, char** argv) {
  const uint32_t S = argc > 1 ? (uint32_t)atoi(argv[1]) : 8192;
  const size_t n = (size_t)S * S;
  float *in, *out;
  hipMalloc(&in, n * 4);
  hipMalloc(&out, n * 4);
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) >> 24);
  hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
  long long* stamps;
  const size_t ns = 8 * 64 * 8;
  hipMalloc(&stamps, ns * 8);
  const bool useStamps = argc > 2 && atoi(argv[2]) != 0;
  g_lab_stamps = useStamps ? stamps : nullptr;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const float sigmas[6] = {0.70710678f, 1.0f, 1.41421356f, 2.0f, 2.82842712f, 4.0f};
  for (int lv = 0; lv < 6; ++lv) {
    float w[129];
    const int taps = ssrlcv_gauss_kernel_host(sigmas[lv], 0.5f, w);
    int ksz = taps | 1;
    hipMemset(stamps, 0, ns * 8);
    launch_conv(in, out, nullptr, S, S, ksz, w, nullptr, nullptr, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch_conv(in, out, nullptr, S, S, ksz, w, nullptr, nullptr, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> st(ns);
    hipMemcpy(st.data(), stamps, ns * 8, hipMemcpyDeviceToHost);
    printf("taps %d: %.3f ms per %ux%u level\n", ksz, ms / 5, S, S);
    for (int wave = 0; wave < 8; ++wave) {
      const int role = (wave ^ (wave >> 2)) & 1;
      double d[6] = {0, 0, 0, 0, 0, 0};
      int cnt = 0;
      for (int it = 8; it < 40; ++it) {
        const long long* s0 = &st[((size_t)wave * 64 + it) * 8];
        const long long* s1 = &st[((size_t)wave * 64 + it + 1) * 8];
        if (!s0[0] || !s1[0]) continue;
        ++cnt;
        if (role == 0) {
          d[0] += s0[1] - s0[0]; d[1] += s0[2] - s0[1]; d[2] += s0[3] - s0[2]; d[3] += s0[4] - s0[3]; d[4] += s1[0] - s0[4];
        } else {
          d[0] += s0[1] - s0[0]; d[1] += s0[2] - s0[1]; d[2] += s1[0] - s0[3]; d[3] += s0[3] - s0[2];
        }
        d[5] += s1[0] - s0[0];
      }
      if (!cnt) continue;
      if (role == 0)
        printf("  wave %d H: mfma %5.0f  ring %5.0f  stage %5.0f  fetch %5.0f  barrier %5.0f | step %5.0f\n", wave, d[0] / cnt,
               d[1] / cnt, d[2] / cnt, d[3] / cnt, d[4] / cnt, d[5] / cnt);
      else
        printf("  wave %d V: mfma %5.0f  store %5.0f  addr %5.0f  barrier %5.0f | step %5.0f\n", wave, d[0] / cnt, d[1] / cnt, d[3] / cnt, d[2] / cnt, d[5] / cnt);
    }
  }
  return 0;
}
