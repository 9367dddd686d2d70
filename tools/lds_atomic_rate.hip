// Developer micro-benchmark: issue cost of LDS atomics per wave-instruction per SIMD (cycles at 2.4 GHz) for
// ds_add_u32 / ds_add_u64 / ds_add_f32, conflict-free or with the address pattern of k_descriptors (8 orientation bins x 4
// lane-private copies: 32 distinct addresses per wave), with all 64 lanes or a quarter of them active.
// build: hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_rate.hip -o tools/_build/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND, int PATTERN, int ACTIVE>
__global__ __launch_bounds__(256) void k_atom(float* out, int iters) {
  __shared__ unsigned long long sh[4][512];
  for (int i = threadIdx.x; i < 2048; i += 256) (&sh[0][0])[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned idx;
  if (PATTERN == 0) idx = lane;                                              // 64 distinct, consecutive
  else if (PATTERN == 1) idx = (((lane * 2654435761u) >> 13) & 7) * 4 + (lane & 3);  // bin * 4 + copy
  else if (PATTERN == 2) idx = lane >> 1;    // neighbours share
  else if (PATTERN == 3) idx = lane & 31;    // lanes l, l + 32 share
  else if (PATTERN == 4) idx = lane & 7;     // 8 lanes per address
  else if (PATTERN == 5) idx = 0;            // one address
  else if (PATTERN == 6) idx = lane & 15;    // 4 lanes per address, one per 16-lane row
  else idx = (((lane * 2654435761u) >> 13) & 7) * 16 + (lane & 15);  // bin * 16 + copy: 16 copies
  unsigned addr = (unsigned)(size_t)(&sh[wave][0]) + idx * 8;
  unsigned long long v64 = lane + 1;
  unsigned v32 = lane + 1;
  float vf = 1.0f;
  const bool on = ACTIVE == 0 || (ACTIVE == 1 && (lane & 3) == 0) || (ACTIVE == 2 && lane < 16);
  if (on) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (KIND == 0) asm volatile("ds_add_u32 %0, %1 offset:%2" : : "v"(addr), "v"(v32), "n"(u * 256) : "memory");
        if (KIND == 1) asm volatile("ds_add_u64 %0, %1 offset:%2" : : "v"(addr), "v"(v64), "n"(u * 256) : "memory");
        if (KIND == 2) asm volatile("ds_add_f32 %0, %1 offset:%2" : : "v"(addr), "v"(vf), "n"(u * 256) : "memory");
      }
    }
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = (float)sh[wave][lane];
}
typedef void (*kern_t)(float*, int);
int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 8192 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  struct E { const char* name; kern_t fn; } tab[] = {
      {"ds_add_u32 conflict-free, 64 lanes", k_atom<0, 0, 0>}, {"ds_add_u64 conflict-free, 64 lanes", k_atom<1, 0, 0>},
      {"ds_add_f32 conflict-free, 64 lanes", k_atom<2, 0, 0>}, {"ds_add_u32 bins x copies,  64 lanes", k_atom<0, 1, 0>},
      {"ds_add_u64 bins x copies,  64 lanes", k_atom<1, 1, 0>}, {"ds_add_u32 conflict-free, lanes%4==0", k_atom<0, 0, 1>},
      {"ds_add_u64 conflict-free, lanes%4==0", k_atom<1, 0, 1>}, {"ds_add_u32 conflict-free, lanes<16", k_atom<0, 0, 2>},
      {"ds_add_u64 conflict-free, lanes<16", k_atom<1, 0, 2>},   {"ds_add_u32 bins x copies,  lanes%4==0", k_atom<0, 1, 1>},
      {"ds_add_u64 bins x copies,  lanes%4==0", k_atom<1, 1, 1>},
      {"ds_add_u64 lane>>1", k_atom<1, 2, 0>}, {"ds_add_u64 lane&31", k_atom<1, 3, 0>}, {"ds_add_u64 lane&7", k_atom<1, 4, 0>},
      {"ds_add_u64 one address", k_atom<1, 5, 0>}, {"ds_add_u64 lane&15", k_atom<1, 6, 0>},
      {"ds_add_u32 lane>>1", k_atom<0, 2, 0>}, {"ds_add_u32 lane&31", k_atom<0, 3, 0>}, {"ds_add_u32 lane&7", k_atom<0, 4, 0>},
      {"ds_add_u32 one address", k_atom<0, 5, 0>}, {"ds_add_u32 lane&15", k_atom<0, 6, 0>},
      {"ds_add_u32 bins x 16 copies", k_atom<0, 7, 0>}, {"ds_add_u64 bins x 16 copies", k_atom<1, 7, 0>}};
  for (const E& e : tab) {
    printf("%-40s", e.name);
    for (int w = 1; w <= 8; w *= 2) {
      int blocks = 256 * w, iters = 2000;
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, 10);
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, iters);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      printf("  w%d: %6.1f", w, ms * 1e-3 * 2.4e9 / ((double)iters * 16 * w));
    }
    printf("  cyc/instr/SIMD\n");
  }
  return 0;
}
