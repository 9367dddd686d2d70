// Developer micro-benchmark: sustained issue cost of individual gfx950 instructions per SIMD at 1/2/4/8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate ; prints ns and cycles (at 2.4 GHz) per
// wave-instruction per SIMD.  Each kernel runs 8 independent chains of one instruction, 16x unrolled.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CHAIN8(OP)            \
  OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)

#define DEF_KERNEL(NAME, TYPE, INIT, OP)                                                   \
  __global__ void NAME(float* out, int iters, float a, float b) {                          \
    TYPE x0 = INIT(0), x1 = INIT(1), x2 = INIT(2), x3 = INIT(3), x4 = INIT(4), x5 = INIT(5), x6 = INIT(6), x7 = INIT(7); \
    for (int i = 0; i < iters; ++i) {                                                      \
      _Pragma("unroll") for (int u = 0; u < 16; ++u) { CHAIN8(OP) }                        \
    }                                                                                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = SUM8;                                     \
  }

#define INITF(k) ((float)threadIdx.x + (k))
#define INITU(k) ((unsigned)threadIdx.x * 2654435761u + (k))
#define INITV(k) (f32x2{(float)threadIdx.x + (k), (float)threadIdx.x - (k)})
#define SUM8 (float)(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7)

#define OP_FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define OP_MUL(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a));
#define OP_ADD(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_CNDMASK(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b));
#define OP_CMP(x) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
#define OP_CMPS(x) asm volatile("v_cmp_le_f32 s[20:21], %0, %1" : : "v"(x), "v"(b) : "s20", "s21");
#define OP_CVTU(x) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(x));
#define OP_EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
#define OP_FRACT(x) asm volatile("v_fract_f32 %0, %0" : "+v"(x));
#define OP_MULHI(x) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(iters));
#define OP_MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(iters));
#define OP_ADDU(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(iters));
#define OP_LSHLADD(x) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x) : "v"(iters));
#define OP_PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(av), "v"(bv));
#define OP_PKMUL(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(av));
#define OP_PKADD(x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(bv));
// VALU followed by an independent SALU / nop / untaken branch from the same wave
#define OP_FMA_SALU(x) asm volatile("v_fma_f32 %0, %0, %1, %2\n s_and_b64 s[20:21], s[20:21], exec" : "+v"(x) : "v"(a), "v"(b) : "s20", "s21", "scc");
#define OP_FMA_NOP(x) asm volatile("v_fma_f32 %0, %0, %1, %2\n s_nop 0" : "+v"(x) : "v"(a), "v"(b));
#define OP_FMA_BR(x) asm volatile("v_fma_f32 %0, %0, %1, %2\n s_cbranch_execz 1f\n1:" : "+v"(x) : "v"(a), "v"(b));
// compare into an SGPR pair, and it, save exec, restore: the per-cell control skeleton of k_descriptors
#define OP_CELL(x) asm volatile("v_cmp_le_f32 s[20:21], %0, %1\n s_and_b64 s[20:21], s[20:21], exec\n s_and_saveexec_b64 s[22:23], s[20:21]\n v_fma_f32 %0, %0, %1, %2\n s_or_b64 exec, exec, s[22:23]" : "+v"(x) : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "scc");

#define OP_CND_E64(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x) : "v"(b));
#define OP_CMP_CND(x) asm volatile("v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc");
#define OP_ADDC(x) asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(x) : : "vcc");
#define OP_MAX(x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_MAXABS(x) asm volatile("v_max_f32_e64 %0, |%0|, |%1|" : "+v"(x) : "v"(b));
#define OP_SUBABS(x) asm volatile("v_sub_f32_e64 %0, 1.0, |%0|" : "+v"(x));
#define OP_MED3(x) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define OP_AND(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(iters));
#define OP_CVTFI(x) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(x));
#define OP_CVTIF(x) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x));
#define OP_FLOOR(x) asm volatile("v_floor_f32 %0, %0" : "+v"(x));
#define OP_RCP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
#define OP_SQRT(x) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x));
#define OP_LDEXP(x) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(x) : "v"(iters));
#define OP_READLANE(x) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(x) : "s20");
#define OP_READFIRST(x) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(x) : "s20");
#define OP_MOV(x) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(b));
#define OP_MOVS(x) asm volatile("v_mov_b32 %0, s20" : "+v"(x));
#define OP_FMA_S(x) asm volatile("v_fma_f32 %0, %0, s20, %1" : "+v"(x) : "v"(b));
#define OP_MUL_S(x) asm volatile("v_mul_f32 %0, s20, %0" : "+v"(x));
#define OP_CMPX(x) asm volatile("s_mov_b64 s[22:23], exec\n v_cmpx_le_f32 vcc, %0, %1\n v_fma_f32 %0, %0, %1, %2\n s_mov_b64 exec, s[22:23]" : "+v"(x) : "v"(a), "v"(b) : "s22", "s23", "vcc");
#define OP_SALU2(x) asm volatile("v_fma_f32 %0, %0, %1, %2\n s_and_b64 s[20:21], s[20:21], exec\n s_or_b64 s[22:23], s[22:23], exec" : "+v"(x) : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "scc");
#define OP_SMOV(x) asm volatile("v_fma_f32 %0, %0, %1, %2\n s_mov_b64 s[20:21], exec" : "+v"(x) : "v"(a), "v"(b) : "s20", "s21");
#define OP_MADI24(x) asm volatile("v_mad_i32_i24 %0, %0, -2, %1" : "+v"(x) : "v"(iters));
#define OP_MIN3I(x) asm volatile("v_min3_i32 %0, %0, %1, %1" : "+v"(x) : "v"(iters));
#define OP_DSADD(x) asm volatile("ds_add_u64 %0, %1" : : "v"(ldsaddr), "v"(x64) : "memory");

DEF_KERNEL(k_fma, float, INITF, OP_FMA)
DEF_KERNEL(k_mul, float, INITF, OP_MUL)
DEF_KERNEL(k_add, float, INITF, OP_ADD)
DEF_KERNEL(k_cndmask, float, INITF, OP_CNDMASK)
DEF_KERNEL(k_cmp, float, INITF, OP_CMP)
DEF_KERNEL(k_cmps, float, INITF, OP_CMPS)
DEF_KERNEL(k_cvtu, float, INITF, OP_CVTU)
DEF_KERNEL(k_exp, float, INITF, OP_EXP)
DEF_KERNEL(k_fract, float, INITF, OP_FRACT)
DEF_KERNEL(k_mulhi, unsigned, INITU, OP_MULHI)
DEF_KERNEL(k_mullo, unsigned, INITU, OP_MULLO)
DEF_KERNEL(k_addu, unsigned, INITU, OP_ADDU)
DEF_KERNEL(k_lshladd, unsigned, INITU, OP_LSHLADD)
DEF_KERNEL(k_fma_salu, float, INITF, OP_FMA_SALU)
DEF_KERNEL(k_fma_nop, float, INITF, OP_FMA_NOP)
DEF_KERNEL(k_fma_br, float, INITF, OP_FMA_BR)
DEF_KERNEL(k_cell, float, INITF, OP_CELL)
DEF_KERNEL(k_madi24, unsigned, INITU, OP_MADI24)
DEF_KERNEL(k_min3i, unsigned, INITU, OP_MIN3I)
DEF_KERNEL(k_cnd64, float, INITF, OP_CND_E64)
DEF_KERNEL(k_cmpcnd, float, INITF, OP_CMP_CND)
DEF_KERNEL(k_addc, unsigned, INITU, OP_ADDC)
DEF_KERNEL(k_max, float, INITF, OP_MAX)
DEF_KERNEL(k_maxabs, float, INITF, OP_MAXABS)
DEF_KERNEL(k_subabs, float, INITF, OP_SUBABS)
DEF_KERNEL(k_med3, float, INITF, OP_MED3)
DEF_KERNEL(k_and, unsigned, INITU, OP_AND)
DEF_KERNEL(k_cvtfi, float, INITF, OP_CVTFI)
DEF_KERNEL(k_cvtif, float, INITF, OP_CVTIF)
DEF_KERNEL(k_floor, float, INITF, OP_FLOOR)
DEF_KERNEL(k_rcp, float, INITF, OP_RCP)
DEF_KERNEL(k_sqrt, float, INITF, OP_SQRT)
DEF_KERNEL(k_ldexp, float, INITF, OP_LDEXP)
DEF_KERNEL(k_readlane, float, INITF, OP_READLANE)
DEF_KERNEL(k_readfirst, float, INITF, OP_READFIRST)
DEF_KERNEL(k_mov, float, INITF, OP_MOV)
DEF_KERNEL(k_movs, float, INITF, OP_MOVS)
DEF_KERNEL(k_fmas, float, INITF, OP_FMA_S)
DEF_KERNEL(k_muls, float, INITF, OP_MUL_S)
DEF_KERNEL(k_cmpx, float, INITF, OP_CMPX)
DEF_KERNEL(k_salu2, float, INITF, OP_SALU2)
DEF_KERNEL(k_smov, float, INITF, OP_SMOV)
__global__ void k_dsadd(float* out, int iters, float a, float b) {
  __shared__ unsigned long long sh[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) sh[i] = 0;
  __syncthreads();
  // 64 distinct addresses per wave, 8 B apart (conflict free), private 512-B region per chain slot
  unsigned ldsaddr = (unsigned)(size_t)(sh + (threadIdx.x >> 6) * 1024 + (threadIdx.x & 63));
  unsigned long long x64 = threadIdx.x;
  float x0 = 0, x1 = 0, x2 = 0, x3 = 0, x4 = 0, x5 = 0, x6 = 0, x7 = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) { CHAIN8(OP_DSADD) }
  }
  __syncthreads();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)sh[threadIdx.x] + SUM8;
}
#undef SUM8
#define SUM8 ((x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7).x)
#define PKPRE f32x2 av = {a, a}, bv = {b, b};
__global__ void k_pkfma(float* out, int iters, float a, float b) {
  PKPRE
  f32x2 x0 = INITV(0), x1 = INITV(1), x2 = INITV(2), x3 = INITV(3), x4 = INITV(4), x5 = INITV(5), x6 = INITV(6), x7 = INITV(7);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) { CHAIN8(OP_PKFMA) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = SUM8;
}
__global__ void k_pkmul(float* out, int iters, float a, float b) {
  PKPRE
  f32x2 x0 = INITV(0), x1 = INITV(1), x2 = INITV(2), x3 = INITV(3), x4 = INITV(4), x5 = INITV(5), x6 = INITV(6), x7 = INITV(7);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) { CHAIN8(OP_PKMUL) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = SUM8 + bv.x;
}
__global__ void k_pkadd(float* out, int iters, float a, float b) {
  PKPRE
  f32x2 x0 = INITV(0), x1 = INITV(1), x2 = INITV(2), x3 = INITV(3), x4 = INITV(4), x5 = INITV(5), x6 = INITV(6), x7 = INITV(7);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) { CHAIN8(OP_PKADD) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = SUM8 + av.x;
}

typedef void (*kern_t)(float*, int, float, float);
struct Entry { const char* name; kern_t fn; };

int main() {
  float* d;
  hipMalloc(&d, 256 * 4096 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  Entry tab[] = {{"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_pk_fma_f32", k_pkfma},
                 {"v_pk_mul_f32", k_pkmul}, {"v_pk_add_f32", k_pkadd}, {"v_cndmask_b32", k_cndmask},
                 {"v_cmp_le_f32 vcc", k_cmp}, {"v_cmp_le_f32 sgpr", k_cmps}, {"v_cvt_u32_f32", k_cvtu},
                 {"v_exp_f32", k_exp}, {"v_fract_f32", k_fract}, {"v_mul_hi_u32", k_mulhi}, {"v_mul_lo_u32", k_mullo},
                 {"v_add_u32", k_addu}, {"v_lshl_add_u32", k_lshladd}, {"v_fma + s_and_b64", k_fma_salu},
                 {"v_fma + s_nop 0", k_fma_nop}, {"v_fma + s_cbranch(untaken)", k_fma_br},
                 {"cell skeleton (cmp,s_and,saveexec,fma,s_or)", k_cell},
                 {"v_cndmask_b32_e64 (sgpr mask)", k_cnd64}, {"v_cmp + v_cndmask vcc", k_cmpcnd}, {"v_addc_co_u32", k_addc},
                 {"v_max_f32", k_max}, {"v_max_f32 |a|,|b|", k_maxabs}, {"v_sub_f32 1.0,|a|", k_subabs},
                 {"v_med3_f32", k_med3}, {"v_and_b32", k_and}, {"v_cvt_f32_i32", k_cvtfi}, {"v_cvt_i32_f32", k_cvtif},
                 {"v_floor_f32", k_floor}, {"v_rcp_f32", k_rcp}, {"v_sqrt_f32", k_sqrt}, {"v_ldexp_f32", k_ldexp},
                 {"v_readlane_b32", k_readlane}, {"v_readfirstlane_b32", k_readfirst}, {"v_mov_b32", k_mov},
                 {"v_mov_b32 v, s", k_movs}, {"v_fma_f32 v,s,v", k_fmas}, {"v_mul_f32 s,v", k_muls},
                 {"s_mov exec; v_cmpx; v_fma; s_mov exec", k_cmpx}, {"v_fma + 2 SALU", k_salu2},
                 {"v_fma + s_mov_b64", k_smov}, {"v_mad_i32_i24", k_madi24}, {"v_min3_i32", k_min3i}, {"ds_add_u64 (conflict-free)", k_dsadd}};
  for (const Entry& e : tab) {
    printf("%-46s", e.name);
    for (int wavesPerSimd = 1; wavesPerSimd <= 8; wavesPerSimd *= 2) {
      int blocks = 256 * wavesPerSimd;  // 256 threads = 4 waves = 1 per SIMD per block
      int iters = 4000;
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0001f, 0.5f);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      double groups_per_simd = (double)iters * 128 * wavesPerSimd;  // instruction groups issued on each SIMD
      printf("  w%d: %5.2f cyc", wavesPerSimd, ms * 1e6 / groups_per_simd * 2.4);
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
