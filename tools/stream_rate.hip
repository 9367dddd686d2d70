// Developer micro-benchmark: HBM streaming rates of the access shapes the pyramid kernels use (MI355X: 8 TB/s spec).
// build: hipcc --offload-arch=gfx950 -O3 tools/stream_rate.hip -o tools/_build/stream_rate
// Shapes: copy (1 read + 1 write stream), read-only, write-only, and the DoG kernel's 6 reads + 5 writes; each with the
// grid bounded to B blocks of 256 threads (grid-stride), U float4 per thread and stream in flight, plain or nontemporal.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride * U) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t j = i + u * stride;
      if (j < n) v[u] = NT ? __builtin_nontemporal_load(in + j) : in[j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t j = i + u * stride;
      if (j < n) { if (NT) __builtin_nontemporal_store(v[u], out + j); else out[j] = v[u]; }
    }
  }
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const f32x4* __restrict__ in, float* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  f32x4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride * U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t j = i + u * stride;
      if (j < n) acc += NT ? __builtin_nontemporal_load(in + j) : in[j];
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.0f;
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_write(f32x4* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  const f32x4 v = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride * U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t j = i + u * stride;
      if (j < n) { if (NT) __builtin_nontemporal_store(v, out + j); else out[j] = v; }
    }
  }
}
struct Dog { const f32x4* in[6]; f32x4* out[5]; };
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_dog_like(Dog d, size_t n, float s) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride * U) {
    f32x4 v[U][6];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t j = i + u * stride;
      if (j < n) {
#pragma unroll
        for (int b = 0; b < 6; ++b) v[u][b] = NT ? __builtin_nontemporal_load(d.in[b] + j) : d.in[b][j];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t j = i + u * stride;
      if (j < n) {
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          const f32x4 r = (v[u][b + 1] - v[u][b]) * s;
          if (NT) __builtin_nontemporal_store(r, d.out[b] + j); else d.out[b][j] = r;
        }
      }
    }
  }
}

// chunked: block b owns the contiguous float4 range [b * 256 * C, (b + 1) * 256 * C); no grid-stride loop
template <int C, bool NT>
__global__ __launch_bounds__(256) void k_dog_chunk(Dog d, size_t n, float s) {
  const size_t base = (size_t)blockIdx.x * 256 * C + threadIdx.x;
#pragma unroll 1
  for (int c = 0; c < C; ++c) {
    const size_t j = base + (size_t)c * 256;
    if (j < n) {
      f32x4 v[6];
#pragma unroll
      for (int b = 0; b < 6; ++b) v[b] = NT ? __builtin_nontemporal_load(d.in[b] + j) : d.in[b][j];
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        const f32x4 r = (v[b + 1] - v[b]) * s;
        if (NT) __builtin_nontemporal_store(r, d.out[b] + j); else d.out[b][j] = r;
      }
    }
  }
}
template <int C, bool NT>
__global__ __launch_bounds__(256) void k_copy_chunk(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n) {
  const size_t base = (size_t)blockIdx.x * 256 * C + threadIdx.x;
#pragma unroll 1
  for (int c = 0; c < C; ++c) {
    const size_t j = base + (size_t)c * 256;
    if (j < n) { if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(in + j), out + j); else out[j] = in[j]; }
  }
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename F>
static float time_ms(F f) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  f();
  CHECK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) {
    CHECK(hipEventRecord(e0));
    f();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  return best;
}

int main() {
  const size_t n = (size_t)8192 * 8192 / 4;  // float4 per 8192^2 level
  const size_t bytes = n * 16;
  f32x4* buf[11];
  for (auto& b : buf) { CHECK(hipMalloc(&b, bytes)); CHECK(hipMemset(b, 0, bytes)); }
  float* sink; CHECK(hipMalloc(&sink, 256));
  Dog d;
  for (int b = 0; b < 6; ++b) d.in[b] = buf[b];
  for (int b = 0; b < 5; ++b) d.out[b] = buf[6 + b];
  const unsigned grids[] = {512, 1024, 2048, 4096, 8192, 65536};
#define ROW(NAME, LAUNCH, BYTES)                                                      \
  for (unsigned g : grids) {                                                          \
    float ms = time_ms([&] { LAUNCH; });                                              \
    printf("%-28s blocks %6u  %.3f ms  %.2f TB/s\n", NAME, g, ms, (BYTES) / ms / 1e9); \
  }
  ROW("copy U1", (k_copy<1, false><<<dim3(g), dim3(256), 0, 0>>>(buf[0], buf[1], n)), 2.0 * bytes)
  ROW("copy U2", (k_copy<2, false><<<dim3(g), dim3(256), 0, 0>>>(buf[0], buf[1], n)), 2.0 * bytes)
  ROW("copy U4", (k_copy<4, false><<<dim3(g), dim3(256), 0, 0>>>(buf[0], buf[1], n)), 2.0 * bytes)
  ROW("copy U8", (k_copy<8, false><<<dim3(g), dim3(256), 0, 0>>>(buf[0], buf[1], n)), 2.0 * bytes)
  ROW("copy U4 nt", (k_copy<4, true><<<dim3(g), dim3(256), 0, 0>>>(buf[0], buf[1], n)), 2.0 * bytes)
  ROW("read U4", (k_read<4, false><<<dim3(g), dim3(256), 0, 0>>>(buf[0], sink, n)), 1.0 * bytes)
  ROW("read U8 nt", (k_read<8, true><<<dim3(g), dim3(256), 0, 0>>>(buf[0], sink, n)), 1.0 * bytes)
  ROW("write U4", (k_write<4, false><<<dim3(g), dim3(256), 0, 0>>>(buf[1], n)), 1.0 * bytes)
  ROW("write U4 nt", (k_write<4, true><<<dim3(g), dim3(256), 0, 0>>>(buf[1], n)), 1.0 * bytes)
  ROW("dog-like 6r5w U1", (k_dog_like<1, false><<<dim3(g), dim3(256), 0, 0>>>(d, n, 0.5f)), 11.0 * bytes)
  ROW("dog-like 6r5w U1 nt", (k_dog_like<1, true><<<dim3(g), dim3(256), 0, 0>>>(d, n, 0.5f)), 11.0 * bytes)
  ROW("dog-like 6r5w U2 nt", (k_dog_like<2, true><<<dim3(g), dim3(256), 0, 0>>>(d, n, 0.5f)), 11.0 * bytes)
  ROW("dog-like 6r5w U2", (k_dog_like<2, false><<<dim3(g), dim3(256), 0, 0>>>(d, n, 0.5f)), 11.0 * bytes)
#define CROW(NAME, KERN, C, ARGS, BYTES)                                               \
  {                                                                                    \
    const unsigned g = (unsigned)((n + 256 * C - 1) / (256 * C));                       \
    float ms = time_ms([&] { KERN<<<dim3(g), dim3(256), 0, 0>>> ARGS; });               \
    printf("%-28s blocks %6u  %.3f ms  %.2f TB/s\n", NAME, g, ms, (BYTES) / ms / 1e9);  \
  }
  CROW("dog chunk 1 nt", (k_dog_chunk<1, true>), 1, (d, n, 0.5f), 11.0 * bytes)
  CROW("dog chunk 2 nt", (k_dog_chunk<2, true>), 2, (d, n, 0.5f), 11.0 * bytes)
  CROW("dog chunk 4 nt", (k_dog_chunk<4, true>), 4, (d, n, 0.5f), 11.0 * bytes)
  CROW("dog chunk 8 nt", (k_dog_chunk<8, true>), 8, (d, n, 0.5f), 11.0 * bytes)
  CROW("dog chunk 16 nt", (k_dog_chunk<16, true>), 16, (d, n, 0.5f), 11.0 * bytes)
  CROW("dog chunk 32 nt", (k_dog_chunk<32, true>), 32, (d, n, 0.5f), 11.0 * bytes)
  CROW("dog chunk 4", (k_dog_chunk<4, false>), 4, (d, n, 0.5f), 11.0 * bytes)
  CROW("dog chunk 16", (k_dog_chunk<16, false>), 16, (d, n, 0.5f), 11.0 * bytes)
  CROW("copy chunk 1 nt", (k_copy_chunk<1, true>), 1, (buf[0], buf[1], n), 2.0 * bytes)
  CROW("copy chunk 4 nt", (k_copy_chunk<4, true>), 4, (buf[0], buf[1], n), 2.0 * bytes)
  CROW("copy chunk 16 nt", (k_copy_chunk<16, true>), 16, (buf[0], buf[1], n), 2.0 * bytes)
  CROW("copy chunk 64 nt", (k_copy_chunk<64, true>), 64, (buf[0], buf[1], n), 2.0 * bytes)
  return 0;
}
