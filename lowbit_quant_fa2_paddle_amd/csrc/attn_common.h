// Device helpers shared by the attention kernels (gfx950).
#pragma once
#include <type_traits>

#include "lbfa_common.h"

namespace lbfa {

// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block of 16-bit elements, delivered column-major (lane i gets column
// i of the 4 rows).  EXEC must be all ones.  Issued behind the compiler's back (inline asm, no memory operand): the intrinsic form
// following a `buffer_load ... lds` makes the compiler wait for vmcnt(0) first - it cannot know that the DMA in flight fills the
// OTHER buffer - which parks every wave in the middle of each tile until the prefetch has landed.  The result is valid only
// after lds_wait_keep() has been given the registers.
template <int OFF>
__device__ __forceinline__ f16x4 lds_read_tr16_raw(unsigned lds_addr) {
  f16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF));
  return v;
}
// Wait until at most N operations issued AFTER the reads that produced these registers are still in flight.  N = the reads
// this wave has requested since (LDS returns in order).  Anything else the compiler has outstanding on the same counter (its own
// LDS reads, scalar loads - which may return out of order) only makes the wait stricter: the count cannot fall to N while one
// of the older reads is pending, because N younger LDS reads stand behind it.
template <int N>
__device__ __forceinline__ void lds_wait_keep(f16x4& a, f16x4& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lds_wait_keep(f16x4& a, f16x4& b, f16x4& c, f16x4& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lds_wait_keep(f16x4& a, f16x4& b, f16x4& c, f16x4& d, f16x4& e, f16x4& f, f16x4& g, f16x4& h) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "n"(N));
}
__device__ __forceinline__ unsigned lds_offset_of(const void* shared_ptr) {
  return (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) const char*)shared_ptr;
}

template <int RB>
__device__ __forceinline__ int kx(int row) {  // K-tile 16-B chunk swizzle for rows of RB bytes (int8: RB = D, fp16: 2 D)
  if constexpr (RB == 64) return (row >> 2) & 3;
  else if constexpr (RB == 128) return (row >> 1) & 7;
  else return row & 15;
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

// ---- reductions over the four lanes l, l + 16, l + 32, l + 48 (one lane per 16-lane row) without the LDS crossbar -----------
// v_permlane16_swap exchanges the odd rows of its first operand with the even rows of the second, v_permlane32_swap the two
// halves: after each step both partners hold both values.  Two VALU instructions per step instead of a ds_bpermute round trip.
__device__ __forceinline__ float rows4_sum(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// "Order keys": the scores of the int8 product are the accumulator bits of kMagic + s, which compare like the integers s as
// signed int32 (-inf = 0xFF800000 sorts below all of them) - integer maxima need no NaN canonicalisation in front and come as
// v_max3_i32.  The fp32 scores of the un-quantised product compare as floats: max(a, b) = med3(a, b, +inf), one v_med3_f32
// without the canonicalising v_max the compiler puts in front of an fmaxf on MFMA output.  (Not inline asm: hipcc pads no
// MFMA -> VALU wait states for an asm statement's operands - a hand-written v_max3_f32 read the score accumulators early.)
template <bool AS_INT>
__device__ __forceinline__ float key_max(float a, float b) {
  if constexpr (AS_INT) {
    const int x = __float_as_int(a), y = __float_as_int(b);
    return __int_as_float(x > y ? x : y);
  } else {
    return __builtin_amdgcn_fmed3f(a, b, __builtin_inff());
  }
}
template <bool AS_INT>
__device__ __forceinline__ float key_max3(float a, float b, float c) {
  return key_max<AS_INT>(key_max<AS_INT>(a, b), c);  // int: selected as v_max3_i32
}
template <bool AS_INT>
__device__ __forceinline__ float rows4_key_max(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = key_max<AS_INT>(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return key_max<AS_INT>(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// Raw buffer resource over [base, base+bytes): out-of-range loads return 0 (hardware bounds check).
// Built from kernel arguments and blockIdx-derived scalars only, so it stays in SGPRs.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0));
}

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [B, E)
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

constexpr float kFp8Offset = 8.807f;  // csrc/qattn/attn_utils.cuh:30: p = exp2(s - m + 8.807) -> p_max = 448
constexpr float kMagic = 12582912.0f;  // 1.5 * 2^23: int32 accumulator bits == float(kMagic + s) for |s| < 2^22
constexpr int kMagicBits = 0x4B400000;
constexpr int kQInt8 = 3;  // QT of the attention kernel: int8 codes; LBFA_F16 / LBFA_BF16 = un-quantised Q and K

// bf16 pairs -> fp16 pairs, in place (a 16-byte chunk = 8 elements)
__device__ __forceinline__ u32x4 bf16x8_to_f16x8(u32x4 val) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float lo = __uint_as_float(val[e] << 16), hi = __uint_as_float(val[e] & 0xffff0000u);
    const f16x2 pk = f16x2{(_Float16)lo, (_Float16)hi};
    val[e] = __builtin_bit_cast(unsigned, pk);
  }
  return val;
}

}  // namespace lbfa
