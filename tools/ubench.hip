// Micro-benchmarks of gfx950 issue rates that drive the attention kernel's design (development tool).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define REP8(X) X X X X X X X X
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

// mode: which instruction stream a wave runs.  role split: waves with (wave_id % nrole == 0) run streamA else streamB
enum { M_FMA = 0, M_EXP, M_CVT, M_MAX3, M_CVTPK, M_PKFMA, M_PKMUL, M_PKADD, M_DOT2, M_MFMA_F16, M_MFMA_I8, M_MIX, M_ADD, M_NOP, M_EXPH, M_PKRTZ, M_MAXI3, M_LDEXP, M_PERM, M_EXP_FMA };

template <int MODE, int NFILL>
__device__ __forceinline__ void body(float (&v)[32], f32x16& acc, i32x16& iacc, f16x8 a, f16x8 b, i32x4 ia, i32x4 ib) {
  if constexpr (MODE == M_FMA) {
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 31]), "v"(v[(i + 2) & 31]));
  } else if constexpr (MODE == M_ADD) {
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 31]));
  } else if constexpr (MODE == M_EXP) {
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
  } else if constexpr (MODE == M_CVT) {
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(v[i]));
  } else if constexpr (MODE == M_MAX3) {
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 31]), "v"(v[(i + 2) & 31]));
  } else if constexpr (MODE == M_CVTPK) {
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 31]));
  } else if constexpr (MODE == M_PKFMA) {
#pragma unroll
    for (int i = 0; i < 32; i += 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double*)&v[i]) : "v"(*(double*)&v[(i + 2) & 31]), "v"(*(double*)&v[(i + 4) & 31]));
  } else if constexpr (MODE == M_PKMUL) {
#pragma unroll
    for (int i = 0; i < 32; i += 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&v[i]) : "v"(*(double*)&v[(i + 2) & 31]));
  } else if constexpr (MODE == M_PKADD) {
#pragma unroll
    for (int i = 0; i < 32; i += 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&v[i]) : "v"(*(double*)&v[(i + 2) & 31]));
  } else if constexpr (MODE == M_DOT2) {
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(v[i]) : "v"(v[(i + 1) & 31]), "v"(v[(i + 2) & 31]));
  } else if constexpr (MODE == M_EXPH) {
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));
  } else if constexpr (MODE == M_PKRTZ) {
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 31]));
  } else if constexpr (MODE == M_MAXI3) {
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 31]), "v"(v[(i + 2) & 31]));
  } else if constexpr (MODE == M_LDEXP) {
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 31]));
  } else if constexpr (MODE == M_PERM) {
#pragma unroll
    for (int i = 0; i < 32; i += 2) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v[i]), "+v"(v[i + 1]));
  } else if constexpr (MODE == M_EXP_FMA) {
    // the softmax core: 1 exp + 3 full-rate ops per element
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i + 8) & 31]) : "v"(v[(i + 9) & 31]), "v"(v[(i + 10) & 31]));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(i + 16) & 31]) : "v"(v[(i + 17) & 31]));
      asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[(i + 24) & 31]) : "v"(v[(i + 25) & 31]));
    }
  } else if constexpr (MODE == M_MFMA_F16) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  } else if constexpr (MODE == M_MFMA_I8) {
#pragma unroll
    for (int i = 0; i < 8; ++i) iacc = __builtin_amdgcn_mfma_i32_32x32x32_i8(ia, ib, iacc, 0, 0, 0);
  } else if constexpr (MODE == M_MIX) {
    // one MFMA followed by NFILL independent v_fma, 8 times
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NFILL; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i * NFILL + k) & 31]) : "v"(v[(i + k + 1) & 31]), "v"(v[(i + k + 2) & 31]));
    }
  }
}

template <int MODE_A, int MODE_B, int NFILL>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters, int split) {
  float v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = 1.0f + threadIdx.x * 1e-6f + i * 1e-7f;
  f32x16 acc = {0};
  i32x16 iacc = {0};
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(i * 0.01f); }
  i32x4 ia = {(int)threadIdx.x, 2, 3, 4}, ib = {5, 6, 7, (int)threadIdx.x};
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // waves are dealt to SIMDs round-robin: waves w and w+4 share a SIMD.  split=1: waves >= 4*? use stream B
  const bool roleB = split && (wave >= 4) && (((wave >> 2) & 1) == 1);
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  if (!roleB) {
    for (int it = 0; it < iters; ++it) body<MODE_A, NFILL>(v, acc, iacc, a, b, ia, ib);
  } else {
    for (int it = 0; it < iters; ++it) body<MODE_B, NFILL>(v, acc, iacc, a, b, ia, ib);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += v[i];
  for (int i = 0; i < 16; ++i) s += acc[i] + (float)iacc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int MA, int MB, int NF>
void run(const char* name, int threads, int split, int ops_a, int ops_b) {
  float* out; long long* cyc;
  const int grid = 256, iters = 2000;
  hipMalloc(&out, grid * 1024 * 4);
  hipMalloc(&cyc, grid * 16 * 8);
  hipMemset(cyc, 0, grid * 16 * 8);
  hipLaunchKernelGGL((k<MA, MB, NF>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters, split);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MA, MB, NF>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters, split);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(grid * 16);
  hipMemcpy(h.data(), cyc, grid * 16 * 8, hipMemcpyDeviceToHost);
  const int nw = threads / 64;
  double ca = 0, cb = 0; int na = 0, nb = 0;
  for (int g = 0; g < grid; ++g) for (int w = 0; w < nw; ++w) {
    bool roleB = split && (w >= 4) && (((w >> 2) & 1) == 1);
    if (roleB) { cb += h[g * 16 + w]; nb++; } else { ca += h[g * 16 + w]; na++; }
  }
  // s_memtime counts at 100 MHz on gfx9? report both raw ticks/iter and wall-derived ns
  printf("%-34s waves/SIMD=%d  A: %8.1f ticks/iter (%6.2f /op)", name, nw / 4, ca / na / iters, ca / na / iters / ops_a);
  if (nb) printf("  B: %8.1f ticks/iter (%6.2f /op)", cb / nb / iters, cb / nb / iters / ops_b);
  printf("  wall %.3f ms -> %.1f ns/iter\n", ms, ms * 1e6 / iters);
  hipFree(out); hipFree(cyc);
}

int main() {
#define SWEEP(M, name, ops) run<M, M, 0>(name, 256, 0, ops, ops); run<M, M, 0>(name, 512, 0, ops, ops); run<M, M, 0>(name, 768, 0, ops, ops); run<M, M, 0>(name, 1024, 0, ops, ops);
  SWEEP(M_FMA, "v_fma_f32 x32", 32)
  SWEEP(M_ADD, "v_add_f32 x32", 32)
  SWEEP(M_PKFMA, "v_pk_fma_f32 x16 (32 elems)", 16)
  SWEEP(M_PKADD, "v_pk_add_f32 x16 (32 elems)", 16)
  SWEEP(M_PKMUL, "v_pk_mul_f32 x16 (32 elems)", 16)
  SWEEP(M_EXP, "v_exp_f32 x32", 32)
  SWEEP(M_EXPH, "v_exp_f16 x32", 32)
  SWEEP(M_PKRTZ, "v_cvt_pkrtz_f16_f32 x32", 32)
  SWEEP(M_MAXI3, "v_max3_i32 x32", 32)
  SWEEP(M_LDEXP, "v_ldexp_f32 x32", 32)
  SWEEP(M_PERM, "v_permlane32_swap x16", 16)
  SWEEP(M_EXP_FMA, "exp+fma+add+sub x32", 32)
  run<M_MFMA_F16, M_EXP_FMA, 0>("A=mfma x8 | B=exp+3 x32", 512, 1, 8, 32);
  run<M_MFMA_F16, M_EXP_FMA, 0>("A=mfma x8 | B=exp+3 x32 (3w: A,B,A)", 768, 1, 8, 32);
  run<M_MFMA_F16, M_EXP_FMA, 0>("A=mfma x8 | B=exp+3 x32 (4w: A,B,A,B)", 1024, 1, 8, 32);
  return 0;
}
