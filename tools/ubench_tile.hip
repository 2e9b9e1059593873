// Synthetic 64x32 score-tile loop of the attention kernel WITHOUT any memory traffic: how many SIMD cycles per wave-tile
// does a given instruction stream cost at 1 / 2 / 3 waves per SIMD?  (development tool; results are garbage numbers)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_tile.hip -o variants/ubench_tile && variants/ubench_tile
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// variants
//  0 serial, VALU row sums (round-1 stream without the overflow check)
//  1 serial, 4x4x4 MFMA row sums
//  2 as 1, PV(block 0) pinned between the exponentials of block 1 (sched_group_barrier)
//  3 software pipeline across tiles: QK(t+1) and PV(t-1) MFMAs spread between the exponentials of tile t (VALU row sums)
//  4 as 3 with 4x4x4 MFMA row sums
//  5 as 3, no row sums at all (lower bound of the pipelined stream)
//  6 MFMAs only (12 per tile)       7 VALU only (32 fma + 32 exp + 16 cvt)
//  8 as 0 and 9 as 3 with the 16x16 MFMA shapes (8 x i32_16x16x64_i8 + 16 x f32_16x16x32_f16 per tile: same FLOPs, same
//    accumulator registers) - does the chip hold a higher clock on them (MI355X_MICROARCH.md, DVFS give-back item 7)?
template <int V>
__global__ __launch_bounds__(256, 2) void tile_loop(float* out, long long* cyc, int iters, float sc, float c1) {
  const int lane = threadIdx.x & 63;
  i32x4 qf[2], kf[4];
  for (int i = 0; i < 2; ++i) qf[i] = i32x4{lane, lane * 3, lane * 5, lane * 7};
  for (int i = 0; i < 4; ++i) kf[i] = i32x4{lane + i, lane * 3 + i, lane * 5 + i, lane * 7 + i};
  f16x8 vf[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) vf[i][e] = (_Float16)(0.001f * (lane + i + e));
  f32x16 acc_o[2] = {{0}, {0}};
  i32x16 cmagic;
  for (int i = 0; i < 16; ++i) cmagic[i] = 0x4B400000;
  asm volatile("" : "+v"(cmagic));
  f32x4 l4 = {0, 0, 0, 0};
  float l = 0.f;
  const f16x4 ones4 = {1, 1, 1, 1};
  float x[2][16];   // scores of the tile being exponentiated
  float xn[2][16];  // scores of the next tile (pipelined variants)
  f16x8 pf[4], pfn[4];
  for (int k = 0; k < 2; ++k)
    for (int i = 0; i < 16; ++i) x[k][i] = xn[k][i] = 12582912.0f + lane + i;
  for (int k = 0; k < 4; ++k)
    for (int e = 0; e < 8; ++e) pf[k][e] = pfn[k][e] = (_Float16)0.5f;

  constexpr bool S16 = (V == 8 || V == 9);
  i32x4 cmagic4 = {0x4B400000, 0x4B400000, 0x4B400000, 0x4B400000};
  asm volatile("" : "+v"(cmagic4));
  f32x4 acc16[8];
  for (int i = 0; i < 8; ++i) acc16[i] = f32x4{0, 0, 0, 0};
  auto qk = [&](float (&dst)[2][16]) __attribute__((always_inline)) {
    if constexpr (S16) {  // 4 key blocks x 2 row blocks of 16x16, K = 64 in one MFMA
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int rbk = 0; rbk < 2; ++rbk) {
          const i32x4 sacc = __builtin_amdgcn_mfma_i32_16x16x64_i8(kf[kb], qf[rbk], cmagic4, 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 4; ++i) dst[kb >> 1][8 * (kb & 1) + 4 * rbk + i] = __int_as_float(sacc[i]);
        }
      return;
    }
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) {
      i32x16 s = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf[2 * kb2], qf[0], cmagic, 0, 0, 0);
      s = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf[2 * kb2 + 1], qf[1], s, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) dst[kb2][i] = __int_as_float(s[i]);
    }
  };
  auto sm = [&](float (&src)[2][16], f16x8 (&dstp)[4], auto kb2_tag, bool vsum) __attribute__((always_inline)) {
    constexpr int kb2 = decltype(kb2_tag)::value;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      src[kb2][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(src[kb2][i], sc, c1));
      if (vsum) l += src[kb2][i];
    }
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int e = 0; e < 8; ++e) dstp[2 * kb2 + g][e] = (_Float16)src[kb2][8 * g + e];
  };
  auto pv = [&](f16x8 (&p)[4], auto kb2_tag, bool msum) __attribute__((always_inline)) {
    constexpr int kb2 = decltype(kb2_tag)::value;
    if constexpr (S16) {  // per 32-key half: 4 channel blocks x 2 row blocks of 16x16x32
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int rbk = 0; rbk < 2; ++rbk)
          acc16[2 * cb + rbk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[cb], p[2 * kb2 + rbk], acc16[2 * cb + rbk], 0, 0, 0);
      return;
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int ks = 2 * kb2 + g;
      acc_o[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[ks], p[ks], acc_o[0], 0, 0, 0);
      acc_o[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[(ks + 1) & 3], p[ks], acc_o[1], 0, 0, 0);
      if (msum) {
        l4 = __builtin_amdgcn_mfma_f32_4x4x4f16(ones4, f16x4{p[ks][0], p[ks][1], p[ks][2], p[ks][3]}, l4, 0, 0, 0);
        l4 = __builtin_amdgcn_mfma_f32_4x4x4f16(ones4, f16x4{p[ks][4], p[ks][5], p[ks][6], p[ks][7]}, l4, 0, 0, 0);
      }
    }
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;

  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) kf[i][0] += 1;  // no score MFMA is loop-invariant
    if constexpr (V == 0 || V == 1 || V == 8) {
      qk(x);
      sm(x, pf, K0{}, V != 1);
      pv(pf, K0{}, V == 1);
      sm(x, pf, K1{}, V != 1);
      pv(pf, K1{}, V == 1);
    } else if constexpr (V == 2) {
      qk(x);
      sm(x, pf, K0{}, false);
      __builtin_amdgcn_sched_barrier(0);
      pv(pf, K0{}, true);
      sm(x, pf, K1{}, false);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      pv(pf, K1{}, true);
    } else if constexpr (V == 3 || V == 4 || V == 5 || V == 9) {
      // tile t: exponentiate x -> pfn ; meanwhile QK of tile t+1 -> xn and PV of tile t-1 from pf
      qk(xn);
      pv(pf, K0{}, V == 4);
      pv(pf, K1{}, V == 4);
      sm(x, pfn, K0{}, V == 3 || V == 9);
      sm(x, pfn, K1{}, V == 3 || V == 9);
      constexpr int NM = (V == 4) ? 20 : (V == 9) ? 24 : 12;
      constexpr int NV = (V == 3 || V == 9) ? 112 / NM : 80 / NM;
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) x[k][i] = xn[k][i];
#pragma unroll
      for (int k = 0; k < 4; ++k) pf[k] = pfn[k];
    } else if constexpr (V == 6) {
      qk(x);
      pv(pf, K0{}, false);
      pv(pf, K1{}, false);
      asm volatile("" ::"v"(x[0][0]), "v"(x[1][5]));
    } else {
      sm(x, pf, K0{}, false);
      sm(x, pf, K1{}, false);
#pragma unroll
      for (int k = 0; k < 4; ++k) asm volatile("" ::"v"(pf[k]));
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) x[k][i] = x[k][i] * 0.f + 12582912.0f;  // keep the exponent argument finite
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = l + l4[0];
  for (int i = 0; i < 16; ++i) s += acc_o[0][i] + acc_o[1][i] + x[0][i] + x[1][i];
  for (int i = 0; i < 8; ++i) s += acc16[i][0] + acc16[i][3];
  for (int k = 0; k < 4; ++k) s += (float)pf[k][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int V>
void run(const char* name) {
  float* out;
  long long* cyc;
  const int iters = 2000;
  for (int occ = 1; occ <= 3; ++occ) {
    const int grid = 256 * occ;
    hipMalloc(&out, grid * 256 * 4);
    hipMalloc(&cyc, grid * 4 * 8);
    hipLaunchKernelGGL((tile_loop<V>), dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1e-4f, -1258.0f);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((tile_loop<V>), dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1e-4f, -1258.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid * 4);
    hipMemcpy(h.data(), cyc, grid * 4 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double wave_cyc = (double)h[h.size() / 2] / iters;  // cycles one wave needs per tile
    printf("%-46s waves/SIMD=%d  wave: %7.1f cyc/tile   wall %.3f ms = %6.1f ns per wave-tile and SIMD   clock %.2f GHz\n", name, occ, wave_cyc,
           ms, ms * 1e6 / (iters * occ), wave_cyc * iters / (ms * 1e6));
    hipFree(out);
    hipFree(cyc);
  }
}

int main() {
  run<0>("0 serial, VALU sums");
  run<1>("1 serial, 4x4x4 sums");
  run<2>("2 PV0 between exps of block 1, 4x4x4");
  run<3>("3 pipelined across tiles, VALU sums");
  run<4>("4 pipelined across tiles, 4x4x4 sums");
  run<5>("5 pipelined across tiles, no sums");
  run<8>("8 serial, VALU sums, 16x16 MFMA shapes");
  run<9>("9 pipelined, VALU sums, 16x16 MFMA shapes");
  run<6>("6 MFMA only (4 i8 + 8 f16)");
  run<7>("7 VALU only (32 fma+exp, 16 cvt)");
  return 0;
}
