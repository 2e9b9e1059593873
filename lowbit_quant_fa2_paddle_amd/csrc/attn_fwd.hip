// Fused low-bit FlashAttention-2 forward for gfx950 (CDNA4): INT8 MFMA for QK^T, FP16 (or FP8) MFMA for PV.
//
// Replaces `_attn_fwd` / `_attn_fwd_inner` of the reference's Triton path
// (src/triton/attn_qk_int8_per_block.py:24-167, ..._causal.py:24-214) and, for fp8 V, the arithmetic of
// csrc/qattn/qk_int_sv_f8_cuda.cu:46-692.  Nothing here is derived from those sources' structure; the
// tiling below is chosen for 64-wide wavefronts and the MFMA register layouts:
//
//  * one workgroup = 4 waves = one 128-row Q block (= one q_scale), each wave owns 32 query rows;
//  * K/V stream through LDS in 64-key tiles (= one k_scale each), double-buffered, staged through
//    registers with 16-byte coalesced loads issued one tile ahead (global loads fly during the MFMAs);
//  * the score product is computed TRANSPOSED, S^T = K Q^T with v_mfma_i32_32x32x32_i8, so that a
//    lane owns ONE query row (column of S^T = lane&31) and 16 keys per 32-key block in registers:
//    row max / row sum are in-lane reductions plus a single v_permlane32_swap across the two halves;
//  * S^T accumulators feed the PV product directly as the B operand of v_mfma_f32_32x32x16_f16
//    (O^T = V^T P^T): no LDS round trip for P.  V^T fragments come from a row-major V tile in LDS via
//    ds_read_b64_tr_b16 (hardware transpose);  O^T keeps the query row on the lane, so the online
//    softmax rescale is a per-lane scalar multiply;
//  * dequantisation (q_scale*k_scale) is folded into the exp2 argument with one v_fma.
//
// LDS images (bank-conflict-free for the access patterns above):
//   K tile  [64 keys][D bytes]   16-B chunk c of row r stored at chunk c ^ kx(r)
//   V tile  [64 keys][D fp16]    64-B chunk c of row r stored at chunk c ^ vx(r)
//   V fp8   [D][64 bytes]        already swizzled in HBM by lbfa_quant_v_fp8 -> linear copy
#include <type_traits>

#include "lbfa_common.h"

namespace lbfa {


typedef __fp16 hf16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) hf16x4* lds_hf16x4_ptr;

// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block of 16-bit elements, delivered
// column-major (lane i gets column i of the 4 rows).  EXEC must be all ones.
__device__ __forceinline__ f16x4 lds_read_tr16(const char* addr) {
  const hf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_hf16x4_ptr)(addr));
  return __builtin_bit_cast(f16x4, v);
}

template <int D>
__device__ __forceinline__ int kx(int row) {  // K-tile chunk swizzle
  if constexpr (D == 64) return (row >> 2) & 3;
  else return (row >> 1) & 7;
}
template <int D>
__device__ __forceinline__ int vx(int row) {  // V-tile 64-B chunk swizzle
  if constexpr (D == 64) return (row >> 1) & 1;
  else return row & 3;
}

__device__ __forceinline__ float half_swap_max(float x) {
  // max over the two 32-lane halves holding the same query row
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_swap_sum(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

constexpr float kFp8Offset = 8.807f;  // csrc/qattn/attn_utils.cuh:30: p = exp2(s - m + 8.807) -> p_max = 448

#ifndef LBFA_MAGIC
#define LBFA_MAGIC 0
#endif
#ifndef LBFA_DOT2
#define LBFA_DOT2 0
#endif
#ifndef LBFA_THR
#define LBFA_THR 8.0f
#endif

template <int D, int VT, int OT, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnParams p) {
  constexpr bool FP8 = (VT == LBFA_E4M3);
  // fp8 P is scaled so that its maximum is 448 = e4m3 max (attn_utils.cuh:30): no headroom to defer
  constexpr float THR = FP8 ? 0.0f : LBFA_THR;
  constexpr int KS = D / 32;                         // int8 k-steps of the score product
  constexpr int DB = D / 32;                         // 32-channel blocks of O^T
  constexpr int KBYTES = 64 * D;                     // K tile
  constexpr int VBYTES = FP8 ? 64 * D : 128 * D;     // V tile
  constexpr int KCH = KBYTES / (256 * 16);           // 16-B chunks per thread
  constexpr int VCH = VBYTES / (256 * 16);
  __shared__ __attribute__((aligned(16))) char smem[2 * (KBYTES + VBYTES)];
  char* const ksm0 = smem;
  char* const vsm0 = smem + 2 * KBYTES;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, hh = lane >> 5;

  // ---- which (batch, head, q-block) -----------------------------------------------------------------
  const unsigned w_id = xcd_remap(blockIdx.x, gridDim.x);
  int qt = (int)(w_id % (unsigned)p.nQ);
  const int bh = (int)(w_id / (unsigned)p.nQ);
  if constexpr (CAUSAL) qt = p.nQ - 1 - qt;  // heaviest q-blocks of a head first
  const int b = bh / p.Hq, h = bh % p.Hq, hk = h / p.group;

  const int row0 = qt * 128 + wave * 32;  // first query row of this wave
  const int qrow = row0 + r;

  // ---- Q fragments (B operand of the int8 MFMA): lane (r, hh) holds bytes [32s+16hh, +16) of its row
  i32x4 qf[KS];
  {
    const int8_t* qp = p.q + (int64_t)b * p.qb + (int64_t)h * p.qh + (int64_t)qrow * p.qs + 16 * hh;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (qrow < p.Sq) qf[s] = *reinterpret_cast<const i32x4*>(qp + 32 * s);
      else qf[s] = i32x4{0, 0, 0, 0};
    }
  }
  const float qsc = p.q_scale[((int64_t)b * p.Hq + h) * p.nQ + qt];
  const float* ksc = p.k_scale + ((int64_t)b * p.Hkv + hk) * p.nK;

  const int8_t* kbase = p.k + (int64_t)b * p.kb + (int64_t)hk * p.kh;
  const char* vbase;
  if constexpr (FP8) vbase = (const char*)p.v + (((int64_t)b * p.Hkv + hk) * p.nK) * (int64_t)(D * 64);
  else vbase = (const char*)p.v + 2 * ((int64_t)b * p.vb + (int64_t)hk * p.vh);

  int n_tiles = p.nK;
  if constexpr (CAUSAL) n_tiles = min(p.nK, 2 * (qt + 1));

  // ---- staging registers ----------------------------------------------------------------------------
  u32x4 kreg[KCH], vreg[VCH];
  auto load_tile = [&](int j) {
    const int n0 = j * 64;
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
      const int c = t + 256 * i, row = c / (D / 16), ch = c % (D / 16);
      kreg[i] = u32x4{0, 0, 0, 0};
      if (n0 + row < p.Sk) kreg[i] = *reinterpret_cast<const u32x4*>(kbase + (int64_t)(n0 + row) * p.ks + ch * 16);
    }
    if constexpr (FP8) {
#pragma unroll
      for (int i = 0; i < VCH; ++i)
        vreg[i] = *reinterpret_cast<const u32x4*>(vbase + (int64_t)j * (D * 64) + (t + 256 * i) * 16);
    } else {
#pragma unroll
      for (int i = 0; i < VCH; ++i) {
        const int c = t + 256 * i, row = c / (D / 8), ch = c % (D / 8);
        vreg[i] = u32x4{0, 0, 0, 0};
        if (n0 + row < p.Sk) vreg[i] = *reinterpret_cast<const u32x4*>(vbase + 2 * ((int64_t)(n0 + row) * p.vs) + ch * 16);
      }
    }
  };
  auto store_tile = [&](int buf) {
    char* ksm = ksm0 + buf * KBYTES;
    char* vsm = vsm0 + buf * VBYTES;
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
      const int c = t + 256 * i, row = c / (D / 16), ch = c % (D / 16);
      *reinterpret_cast<u32x4*>(ksm + row * D + ((ch ^ kx<D>(row)) << 4)) = kreg[i];
    }
    if constexpr (FP8) {
#pragma unroll
      for (int i = 0; i < VCH; ++i) *reinterpret_cast<u32x4*>(vsm + (t + 256 * i) * 16) = vreg[i];
    } else {
#pragma unroll
      for (int i = 0; i < VCH; ++i) {
        const int c = t + 256 * i, row = c / (D / 8), ch = c % (D / 8);
        u32x4 val = vreg[i];
        if constexpr (VT == LBFA_BF16) {  // bf16 -> fp16 on the way in (src/core.py:307-308 `v.to(float16)`)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(val[e] << 16), hi = __uint_as_float(val[e] & 0xffff0000u);
            const f16x2 pk = f16x2{(_Float16)lo, (_Float16)hi};
            val[e] = __builtin_bit_cast(unsigned, pk);
          }
        }
        *reinterpret_cast<u32x4*>(vsm + row * (2 * D) + (((ch >> 2) ^ vx<D>(row)) << 6) + ((ch & 3) << 4)) = val;
      }
    }
  };

  // ---- running state --------------------------------------------------------------------------------
  f32x16 acc_o[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc_o[db][i] = 0.f;
  float m_run = -INFINITY;  // running max (base-2 domain), identical in both halves of a row
  float l_run = 0.f;        // running sum over THIS lane's keys only (halves are added in the epilogue)

#if LBFA_MAGIC
  // int32 -> fp32 without a convert: accumulate on top of 1.5*2^23 so the accumulator's BITS are the float
  // 12582912 + s (exact for |s| < 2^22; |s| <= 127*127*128 < 2^21).  The bias is folded into the fma constant.
  i32x16 cmagic;
#pragma unroll
  for (int i = 0; i < 16; ++i) cmagic[i] = 0x4B400000;
  constexpr float kMagic = 12582912.0f;
#endif

  auto compute_tile = [&](int buf, int j, auto masked_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const char* ksm = ksm0 + buf * KBYTES;
    const char* vsm = vsm0 + buf * VBYTES;
    const int n0 = j * 64;
    // -- S^T = K Q^T (int8 -> int32): two 32-key blocks
    i32x16 sacc[2];
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) {
      const int krow = 32 * kb2 + r;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const i32x4 kf = *reinterpret_cast<const i32x4*>(ksm + krow * D + (((2 * s + hh) ^ kx<D>(krow)) << 4));
        if (s == 0) {
#if LBFA_MAGIC
          sacc[kb2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf, qf[s], cmagic, 0, 0, 0);
#else
          sacc[kb2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf, qf[s], i32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0);
#endif
        } else {
          sacc[kb2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf, qf[s], sacc[kb2], 0, 0, 0);
        }
      }
    }
    // -- online softmax, base 2; dequant scale folded into the exponent argument
    const float sc = qsc * ksc[j];
    float x[2][16];
    float mloc = -INFINITY;
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
#if LBFA_MAGIC
        float v = __int_as_float(sacc[kb2][i]);
#else
        float v = (float)sacc[kb2][i];
#endif
        if constexpr (MASKED) {
          const int key = n0 + 32 * kb2 + (i & 3) + 8 * (i >> 2) + 4 * hh;
          bool dead = key >= p.Sk;
          if constexpr (CAUSAL) dead = dead || (key > qrow);
          if (dead) v = -INFINITY;
        }
        x[kb2][i] = v;
        mloc = fmaxf(mloc, v);
      }
    mloc = half_swap_max(mloc);
    // sc > 0, so max commutes with the scaling; -inf stays -inf
#if LBFA_MAGIC
    const float m_cand = fmaxf(m_run, __builtin_fmaf(mloc, sc, -kMagic * sc));
#else
    const float m_cand = fmaxf(m_run, mloc * sc);
#endif
    // Deferred rescale: keep the old reference max while no row of the wave grew by more than THR
    // (P then stays <= 2^THR, exact in fp16/fp32); rescale O and l only when some row did.
    // First tile: m_run = -inf, so the branch is taken and alpha = 0.
    if (__any(m_cand > m_run + THR)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_cand);  // m_run = -inf -> 0
      m_run = m_cand;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc_o[db][i] *= alpha;
    }
#if LBFA_MAGIC
    float cexp = __builtin_fmaf(-kMagic, sc, -m_run);
#else
    float cexp = -m_run;
#endif
    if constexpr (FP8) cexp += kFp8Offset;
    float psum = 0.f;
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(x[kb2][i], sc, cexp));
        x[kb2][i] = pv;
#if !LBFA_DOT2
        psum += pv;
#endif
      }

    // -- O^T += V^T P^T : P^T fragments straight from the score accumulators
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kb2 = ks >> 1, rb = (ks & 1) * 8;
      if constexpr (FP8) {
#if LBFA_DOT2
#pragma unroll
        for (int e = 0; e < 8; ++e) psum += x[kb2][rb + e];
#endif
        unsigned w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(x[kb2][rb + 0], x[kb2][rb + 1], w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(x[kb2][rb + 2], x[kb2][rb + 3], w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(x[kb2][rb + 4], x[kb2][rb + 5], w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(x[kb2][rb + 6], x[kb2][rb + 7], w1, true);
        const long pf = (long)(((unsigned long)w1 << 32) | (unsigned long)w0);
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          const int d = 32 * db + r;
          const long vf = *reinterpret_cast<const long*>(vsm + d * 64 + (((2 * ks + hh) ^ ((d >> 2) & 7)) << 3));
          acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(vf, pf, acc_o[db], 0, 0, 0);
        }
      } else {
        f16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[e] = (_Float16)x[kb2][rb + e];
#if LBFA_DOT2
        // row sum of the fp16-rounded probabilities, two per instruction (v_dot2_f32_f16, fp32 accumulate)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          psum = __builtin_amdgcn_fdot2(f16x2{pf[2 * e], pf[2 * e + 1]}, f16x2{(_Float16)1.0f, (_Float16)1.0f}, psum, false);
#endif
        const int vrow = 16 * ks + 4 * hh + ((lane & 15) >> 2);
        const int vcol = 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          const char* a0 = vsm + vrow * (2 * D) + ((db ^ vx<D>(vrow)) << 6) + vcol;
          const char* a1 = vsm + (vrow + 8) * (2 * D) + ((db ^ vx<D>(vrow + 8)) << 6) + vcol;
          const f16x4 lo = lds_read_tr16(a0);
          const f16x4 hi = lds_read_tr16(a1);
          const f16x8 vf = f16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, acc_o[db], 0, 0, 0);
        }
      }
    }
    l_run += psum;
  };

  // ---- tile loop: one barrier per tile.  Full (unmasked) tiles first, branch-free; then the at most
  // three tiles that need masking (causal diagonal block = 2 tiles, ragged last tile).
  int n_main = n_tiles;
  if constexpr (CAUSAL) n_main = min(n_tiles, 2 * qt);
  else if ((p.Sk & 63) != 0) n_main = n_tiles - 1;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  int j = 0;
  for (; j < n_main; ++j) {
    const int buf = j & 1;
    if (j + 1 < n_tiles) load_tile(j + 1);
    compute_tile(buf, j, std::false_type{});
    if (j + 1 < n_tiles) store_tile(buf ^ 1);
    __syncthreads();
  }
  for (; j < n_tiles; ++j) {
    const int buf = j & 1;
    if (j + 1 < n_tiles) load_tile(j + 1);
    bool skip = false;
    if constexpr (CAUSAL) skip = j * 64 > row0 + 31;  // every key of the tile is above every row of this wave
    if (!skip) compute_tile(buf, j, std::true_type{});
    if (j + 1 < n_tiles) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: O = O^T / l (x v_scale), LSE ------------------------------------------------------------
  const float l_tot = half_swap_sum(l_run);
  const float inv_l = 1.0f / l_tot;
  if (qrow < p.Sq) {
    unsigned short* op = reinterpret_cast<unsigned short*>(p.o) + (int64_t)b * p.ob + (int64_t)h * p.oh + (int64_t)qrow * p.os;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int d0 = 32 * db + 8 * g4 + 4 * hh;
        float o4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = acc_o[db][4 * g4 + e] * inv_l;
        if constexpr (FP8) {
          const f32x4 vs4 = *reinterpret_cast<const f32x4*>(p.v_scale + ((int64_t)b * p.Hkv + hk) * D + d0);
#pragma unroll
          for (int e = 0; e < 4; ++e) o4[e] *= vs4[e];
        }
        uint2 pk;
        pk.x = (unsigned)store_cvt<OT>(o4[0]) | ((unsigned)store_cvt<OT>(o4[1]) << 16);
        pk.y = (unsigned)store_cvt<OT>(o4[2]) | ((unsigned)store_cvt<OT>(o4[3]) << 16);
        *reinterpret_cast<uint2*>(op + d0) = pk;
      }
    if (p.lse != nullptr && hh == 0) {
      float ls = log2f(l_tot) + m_run;  // base-2 domain (attn_qk_int8_per_block.py:164-167)
      if constexpr (FP8) ls -= kFp8Offset;  // qk_int_sv_f8_cuda.cu:689
      p.lse[((int64_t)b * p.Hq + h) * p.Sq + qrow] = ls;
    }
  }
}

hipError_t launch_attn_fwd(const AttnParams& p, int D, int v_dtype, int o_dtype, int causal, hipStream_t stream) {
  const unsigned n = (unsigned)p.B * p.Hq * p.nQ;
  dim3 grid(n), block(256);
#define LBFA_A(DD, VT, OT)                                                                        \
  do {                                                                                            \
    if (causal) hipLaunchKernelGGL((attn_fwd_kernel<DD, VT, OT, true>), grid, block, 0, stream, p);  \
    else hipLaunchKernelGGL((attn_fwd_kernel<DD, VT, OT, false>), grid, block, 0, stream, p);        \
  } while (0)
#define LBFA_A2(DD, VT)                                   \
  do {                                                    \
    if (o_dtype == LBFA_F16) LBFA_A(DD, VT, LBFA_F16);    \
    else LBFA_A(DD, VT, LBFA_BF16);                       \
  } while (0)
#define LBFA_A3(DD)                                       \
  do {                                                    \
    if (v_dtype == LBFA_F16) LBFA_A2(DD, LBFA_F16);       \
    else if (v_dtype == LBFA_BF16) LBFA_A2(DD, LBFA_BF16);\
    else LBFA_A2(DD, LBFA_E4M3);                          \
  } while (0)
  if (D == 64) LBFA_A3(64);
  else LBFA_A3(128);
#undef LBFA_A3
#undef LBFA_A2
#undef LBFA_A
  return hipGetLastError();
}

}  // namespace lbfa
