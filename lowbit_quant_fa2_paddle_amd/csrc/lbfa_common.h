// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of liblowbit_fa_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/lowbit_fa.h"

namespace lbfa {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;

// ---- storage-dtype conversions (raw 16-bit patterns <-> fp32) -------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  // round-to-nearest-even; NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float f16_bits_to_f32(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
__device__ __forceinline__ unsigned short f32_to_f16_bits(float f) {
  _Float16 h = (_Float16)f;
  return __builtin_bit_cast(unsigned short, h);
}
template <int DT>
__device__ __forceinline__ float load_cvt(unsigned short b) {
  if constexpr (DT == LBFA_F16) return f16_bits_to_f32(b);
  else return bf16_bits_to_f32(b);
}
template <int DT>
__device__ __forceinline__ unsigned short store_cvt(float f) {
  if constexpr (DT == LBFA_F16) return f32_to_f16_bits(f);
  else return f32_to_bf16_bits(f);
}

// ---- wave-level reductions ------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// max over the wave of NON-NEGATIVE finite floats (no NaN) without the LDS crossbar: four DPP steps inside each row of 16
// lanes, v_permlane32_swap across the halves, row_bcast:15 across neighbouring rows; lane 63 then holds the maximum.
// Returned wave-uniform (SGPR).  ~8 VALU instructions instead of 6 ds_bpermute round trips.
__device__ __forceinline__ float wave_max_nonneg(float v) {
  auto dpp_max = [](float x, auto ctrl) {
    const int y = __builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, true);  // invalid lanes read 0
    return fmaxf(x, __int_as_float(y));
  };
  v = dpp_max(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
  v = dpp_max(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
  v = dpp_max(v, std::integral_constant<int, 0x141>{});  // row_half_mirror
  v = dpp_max(v, std::integral_constant<int, 0x140>{});  // row_mirror: every lane of a row holds the row's max
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));  // rows 0|2 and 1|3 combined
  v = dpp_max(v, std::integral_constant<int, 0x142>{});     // row_bcast:15 - row k takes lane 15 of row k - 1
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 63));
}

// XCD-aware bijective remap of a 1-D grid: hardware deals consecutive workgroup ids round-robin over
// the 8 XCDs, so ids {x, x+8, x+16, ...} share one L2.  Give each such residue class a CONTIGUOUS
// chunk of the logical work list, so that neighbours in the work list (same (batch, kv-head): same
// K/V panel) hit one XCD's L2.  Speed only - any placement is correct.
__device__ __forceinline__ unsigned xcd_remap(unsigned id, unsigned n) {
  const unsigned xcd = id & 7u, j = id >> 3;
  const unsigned q = n >> 3, r = n & 7u;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + j;
}

// ---- launch parameter blocks shared by the kernels and the C-ABI layer ---------------------------
struct QuantParams {
  const unsigned short* x;
  const unsigned short* mean;        // [B, H/mean_group, D] or null
  int8_t* out;
  float* scale;                      // [B,H,nblk]
  const unsigned short* rowdot_vec;  // [B, H/rowdot_group, D] or null
  float* rowdot_out;                 // [B,H,S]
  int64_t xb, xh, xs, ob, oh, os;
  float sm_scale, qmax;
  int B, H, S, nblk, mean_group, rowdot_group;
  // scale index = base(b) + h * scale_h + blk * scale_blk, base(b) = (cu_scale ? cu_scale[b] : b) * scale_b.
  // Dense [B,H,nblk]: (H*nblk, nblk, 1).  The reference's packed layout [sum_blocks, H]
  // (src/triton/quant_per_block_varlen.py:60,101-106): (H, 1, H) with cu_scale.
  int64_t scale_b, scale_h, scale_blk;
  // Packed variable-length batch (src/core.py:356-491): sequence b owns tokens [cu_seqlens[b], cu_seqlens[b+1]) of
  // x / out (batch strides unused), blocks restart at each sequence.  null = dense.
  const int* cu_seqlens;
  const int* cu_scale;
  int d_valid;  // columns >= d_valid of x are padding: not read, codes 0 (head-dim pad of src/core.py:277-287 without a copy)
  int mean_b;  // 1: mean / rowdot_vec are per batch entry; 0: one vector set shared by all (varlen k.mean(dim=0), :453)
  // Fused last step of the mean (lbfa_forward): instead of reading `mean`, every workgroup adds the fp64 partial sums of its
  // (batch, head) itself - [.., mean_nsplit, D] doubles written by mean_partial_kernel, same order and roundings as
  // mean_finalize_kernel - and the workgroup of block 0 stores the result to `mean_out` (for the LSE correction q . km).
  const double* mean_partial;  // null: read `mean`
  unsigned short* mean_out;    // may be null
  int mean_nsplit, mean_S;
};

struct AttnParams {
  const int8_t* q;
  const int8_t* k;
  const void* v;
  void* o;
  float* lse;
  const float* q_scale;
  const float* k_scale;
  const float* v_scale;
  int64_t qb, qh, qs, kb, kh, ks, vb, vh, vs, ob, oh, os;
  int B, Hq, Hkv, Sq, Sk, nQ, nK, group;
  // LSE post-processing (src/core.py:344-350): lse_out = (log2(l) + m) * lse_scale + lse_corr[b,h,s] * lse_corr_scale.
  // lbfa_attn_fwd uses (1, null): the raw base-2 value the reference kernel stores.
  const float* lse_corr;
  float lse_scale, lse_corr_scale;
  // scale addressing as in QuantParams; packed variable-length batch when cu_q != null (then cu_k too):
  // (src/triton/attn_qk_int8_block_varlen.py:125-160), Sq / Sk / nQ / nK hold the maxima over the batch.
  int64_t qsc_b, qsc_h, qsc_blk, ksc_b, ksc_h, ksc_blk;
  const int* cu_q;
  const int* cu_k;
  const int* cu_qscale;
  const int* cu_kscale;
  int d_valid;  // channels >= d_valid of V / O are padding: V reads as 0, O is not written (o has d_valid columns)
  float qk_scale;  // un-quantised Q / K only: sm_scale * log2(e), applied to the fp32 scores
  // Q quantised inside the attention kernel (one-call operators): q points at the fp16 / bf16 source (dtype = o's, strides in
  // elements), each workgroup quantises its own 128-row block exactly as quant_per_block_kernel does (same codes and scale),
  // and the LSE correction q . km comes from the same registers (q_dot_vec = km [B,Hkv,D] or null).
  float q_sm_scale, q_qmax;
  const unsigned short* q_dot_vec;
};

}  // namespace lbfa
