// Quantisation pre-passes for the low-bit attention path (gfx950).
//   mean_partial / mean_finalize : km = mean_S(K)                         (src/core.py:292-293)
//   quant_per_block              : K - km, * sm_scale, per-block amax, int8 (src/triton/quant_per_block.py:132-178)
//   quant_v_fp8                  : per-channel e4m3 V                      (csrc/fused/fused.cu:317-428)
// All of these are pure HBM streaming: 16-byte loads per lane, values kept in registers between the
// amax pass and the encode pass so every input byte is read exactly once.
// Built with -ffp-contract=off and correctly rounded fp32 division: the int8 codes and the scales are
// bit-exact against the CPU oracle.
#include <type_traits>

#include "lbfa_common.h"

namespace lbfa {

// ---------------------------------------------------------------------------------------------------
// mean over the sequence
// ---------------------------------------------------------------------------------------------------
struct MeanParams {
  const unsigned short* x;
  double* partial;  // [B,H,nsplit,D] - fp64 sums: the mean is the correctly rounded one (== a float64 reference mean), not
                    // "within an ulp": with 4-bit-range codes a 1-ulp shift of km flips codes
  void* out;        // [B,H,D] storage dtype
  int64_t sb, sh, ss;
  int B, H, S, D, rows_per_split, nsplit;
  int d_valid;  // columns >= d_valid are padding: not read, mean 0 (head-dim pad of src/core.py:277-287 without a copy)
};

template <int DT, int D>
__global__ __launch_bounds__(256) void mean_partial_kernel(MeanParams p) {
  constexpr int CPR = D / 8;        // 16-byte chunks per row
  constexpr int RL = 256 / CPR;     // rows in flight per pass
  __shared__ double red[RL][D + 1];
  const int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int t = threadIdx.x, c = t % CPR, rl = t / CPR;
  const unsigned short* base = p.x + (int64_t)b * p.sb + (int64_t)h * p.sh + c * 8;
  const int r0 = split * p.rows_per_split;
  const int r1 = min(r0 + p.rows_per_split, p.S);
  double acc[8];  // the kernel is HBM-bound: fp64 adds are free here
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.0;
  if (c * 8 < p.d_valid)
  for (int r = r0 + rl; r < r1; r += 8 * RL) {
    // eight independent 16-byte loads in flight per lane, then the adds (a load-add-load chain left this kernel
    // latency-bound: 31.8 -> 28.5 us at D = 128, 13.0 -> 12.4 us at C2)
    uint4 raw[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      raw[u] = make_uint4(0, 0, 0, 0);
      if (r + u * RL < r1) raw[u] = *reinterpret_cast<const uint4*>(base + (int64_t)(r + u * RL) * p.ss);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned w[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[2 * i] += (double)load_cvt<DT>((unsigned short)(w[i] & 0xffffu));
        acc[2 * i + 1] += (double)load_cvt<DT>((unsigned short)(w[i] >> 16));
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[rl][c * 8 + i] = acc[i];
  __syncthreads();
  if (t < D) {
    double s = 0.0;
#pragma unroll 4
    for (int r = 0; r < RL; ++r) s += red[r][t];
    p.partial[(((int64_t)b * p.H + h) * p.nsplit + split) * D + t] = s;
  }
}

template <int DT>
__global__ void mean_finalize_kernel(MeanParams p) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over B*H*D
  const int64_t n = (int64_t)p.B * p.H * p.D;
  if (idx >= n) return;
  const int d = (int)(idx % p.D);
  const int64_t bh = idx / p.D;
  const double* src = p.partial + bh * p.nsplit * p.D + d;
  // loads issued 16 at a time (a dependent chain of single loads made this tiny kernel latency-bound), added in split order
  double s = 0.0;
  for (int i0 = 0; i0 < p.nsplit; i0 += 16) {
    double v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = (i0 + j < p.nsplit) ? src[(int64_t)(i0 + j) * p.D] : 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (i0 + j < p.nsplit) s += v[j];
  }
  // float64 mean -> fp32 -> storage dtype: the roundings of `np.mean(float64).astype(float32)` -> fp16 / bf16
  reinterpret_cast<unsigned short*>(p.out)[idx] = store_cvt<DT>((float)(s / (double)p.S));
}

// ---------------------------------------------------------------------------------------------------
// per-block int8 / int4-range quantiser
// ---------------------------------------------------------------------------------------------------

template <int DT>
__device__ __forceinline__ void unpack8(const uint4& raw, float (&v)[8]) {
  const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = load_cvt<DT>((unsigned short)(w[i] & 0xffffu));
    v[2 * i + 1] = load_cvt<DT>((unsigned short)(w[i] >> 16));
  }
}

// NB = consecutive blocks of BLK rows handled by one workgroup (all their rows are requested before any arithmetic; the
// mean / dot vectors and the fused last step of the mean are fetched once for all of them).
template <int DT, int D, int BLK, bool HAS_MEAN, bool HAS_DOT, int NB = 1>
__global__ __launch_bounds__(256) void quant_per_block_kernel(QuantParams p) {
  constexpr int CPR = D / 8;
  constexpr int RPP = 256 / CPR;  // rows per pass
  constexpr int NP = BLK / RPP;   // passes per block
  __shared__ float wmax[NB][4];
  __shared__ float smean[D];
  const int blk0 = blockIdx.x * NB, h = blockIdx.y, b = blockIdx.z;
  const int t = threadIdx.x, c = t % CPR, rl = t / CPR;
  int S = p.S;
  int64_t xoff = (int64_t)b * p.xb, ooff = (int64_t)b * p.ob, sbase = (int64_t)b * p.scale_b;
  if (p.cu_seqlens != nullptr) {  // packed batch: sequence b (quant_per_block_varlen.py:41-48)
    const int s0 = p.cu_seqlens[b];
    S = p.cu_seqlens[b + 1] - s0;
    if (blk0 * BLK >= S) return;
    xoff = (int64_t)s0 * p.xs;
    ooff = (int64_t)s0 * p.os;
    if (p.cu_scale != nullptr) sbase = (int64_t)p.cu_scale[b] * p.scale_b;
  }
  const unsigned short* xbase = p.x + xoff + (int64_t)h * p.xh + c * 8;
  int8_t* obase = p.out + ooff + (int64_t)h * p.oh + c * 8;
  const int vb = b * p.mean_b;  // which mean / rowdot vector set

  float mean[8], vec[8];
  const bool fused_mean = HAS_MEAN && p.mean_partial != nullptr;  // kernel-uniform
  if constexpr (HAS_MEAN) {
    if (!fused_mean)
      unpack8<DT>(*reinterpret_cast<const uint4*>(p.mean + ((int64_t)vb * (p.H / p.mean_group) + h / p.mean_group) * D + c * 8), mean);
  }
  if constexpr (HAS_DOT)
    unpack8<DT>(*reinterpret_cast<const uint4*>(p.rowdot_vec + ((int64_t)vb * (p.H / p.rowdot_group) + h / p.rowdot_group) * D + c * 8), vec);

  // all loads first (NB * NP independent 16-byte loads in flight per lane), then the arithmetic
  uint4 raw[NB][NP];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int row = (blk0 + nb) * BLK + ps * RPP + rl;
      raw[nb][ps] = make_uint4(0, 0, 0, 0);  // masked rows load as 0 (quant_per_block.py:170)
      if (row < S && c * 8 < p.d_valid) raw[nb][ps] = *reinterpret_cast<const uint4*>(xbase + (int64_t)row * p.xs);
    }
  if constexpr (HAS_MEAN) {
    if (fused_mean) {  // last step of the mean, while the rows are in flight: one lane per channel
      if (t < D) {
        const int64_t bh = (int64_t)vb * (p.H / p.mean_group) + h / p.mean_group;
        const double* src = p.mean_partial + bh * p.mean_nsplit * D + t;
        double s = 0.0;  // loads issued 16 at a time, added in split order: exactly mean_finalize_kernel
        for (int i0 = 0; i0 < p.mean_nsplit; i0 += 16) {
          double v[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] = (i0 + j < p.mean_nsplit) ? src[(int64_t)(i0 + j) * D] : 0.0;
#pragma unroll
          for (int j = 0; j < 16; ++j)
            if (i0 + j < p.mean_nsplit) s += v[j];
        }
        const unsigned short km = store_cvt<DT>((float)(s / (double)p.mean_S));
        smean[t] = load_cvt<DT>(km);
        if (blk0 == 0 && p.mean_out != nullptr) p.mean_out[bh * D + t] = km;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; ++i) mean[i] = smean[c * 8 + i];
    }
  }
  float xs[NB][NP][8];
  float amax[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    amax[nb] = 0.f;
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int row = (blk0 + nb) * BLK + ps * RPP + rl;
      float v[8];
      unpack8<DT>(raw[nb][ps], v);
      if constexpr (HAS_DOT) {
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) dot += v[i] * vec[i];
#pragma unroll
        for (int o = CPR / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
        if (c == 0 && row < S) p.rowdot_out[((int64_t)b * p.H + h) * S + row] = load_cvt<DT>(store_cvt<DT>(dot));
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float x = v[i];
        // `k - km` is an elementwise op in the storage dtype (quant_per_block.py:186-187): the fp32 difference
        // is rounded to that dtype, as the CPU oracle (and torch/paddle CPU) does.  Rows past the end stay 0:
        // the reference subtracts on the real tensor, then loads masked rows as 0.
        if constexpr (HAS_MEAN) x = (row < S) ? load_cvt<DT>(store_cvt<DT>(x - mean[i])) : 0.f;
        x *= p.sm_scale;
        xs[nb][ps][i] = x;
        amax[nb] = fmaxf(amax[nb], fabsf(x));
      }
    }
    amax[nb] = wave_max_nonneg(amax[nb]);
    if ((t & 63) == 0) wmax[nb][t >> 6] = amax[nb];
  }
  __syncthreads();

  // y = xs / scale must be the correctly rounded fp32 quotient (the codes are bit-exact against the oracle).
  // A full IEEE division per element costs ~12 VALU ops; the divisor is the same for the whole block, so use
  // Markstein's sequence with the correctly rounded reciprocal: q0 = x*r, e = fma(-q0, s, x) (exact),
  // q1 = fma(e, r, q0) == RN(x/s) for every x unless the significand of s is all ones (checked on 1.2e8
  // adversarial samples; theorem: Markstein 1990), in which case the plain division is used.
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int blk = blk0 + nb;
    if (blk * BLK >= S) break;  // workgroup-uniform: a block past the end of the sequence owns no scale slot
    const float am = fmaxf(fmaxf(wmax[nb][0], wmax[nb][1]), fmaxf(wmax[nb][2], wmax[nb][3]));
    const float scale = fmaxf(am, 1e-7f) / p.qmax;
    if (t == 0) p.scale[sbase + (int64_t)h * p.scale_h + (int64_t)blk * p.scale_blk] = scale;
    const float rcp = 1.0f / scale;
    const bool exact_rcp_ok =
        (__builtin_amdgcn_readfirstlane(__float_as_uint(scale)) & 0x7fffffu) != 0x7fffffu;  // block-uniform
    auto encode = [&](auto fast_tag) {
      constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) {
        const int row = blk * BLK + ps * RPP + rl;
        int q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xv = xs[nb][ps][i];
          float y;
          if constexpr (FAST) {
            const float q0 = xv * rcp;
            y = __builtin_fmaf(__builtin_fmaf(-q0, scale, xv), rcp, q0);
          } else {
            y = xv / scale;
          }
          // round half away from zero (:174-176): y + 0.5*sign(y), then truncate (v_cvt_i32_f32 truncates)
          q[i] = (int)(y + __builtin_copysignf(0.5f, y));
        }
        // |q| <= 127: v_cvt_pk_i16_i32 keeps the low bytes, v_perm_b32 gathers bytes 0 and 2 of each pair
        const unsigned p01 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pk_i16(q[0], q[1]));
        const unsigned p23 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pk_i16(q[2], q[3]));
        const unsigned p45 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pk_i16(q[4], q[5]));
        const unsigned p67 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pk_i16(q[6], q[7]));
        const unsigned w0 = __builtin_amdgcn_perm(p23, p01, 0x06040200u);
        const unsigned w1 = __builtin_amdgcn_perm(p67, p45, 0x06040200u);
        if (row < S) *reinterpret_cast<uint2*>(obase + (int64_t)row * p.os) = make_uint2(w0, w1);
      }
    };
    if (exact_rcp_ok) encode(std::true_type{});
    else encode(std::false_type{});
  }
}

// ---------------------------------------------------------------------------------------------------
// per-channel e4m3 quantiser for V
// ---------------------------------------------------------------------------------------------------
struct VFp8Params {
  const unsigned short* v;
  uint8_t* out;        // [B,H,ntile,D,64]
  float* v_scale;      // [B,H,D]
  unsigned* amax_bits; // [B,H,D] scratch at the tail of the v_fp8 allocation; zeroed before the launch
  int64_t vb, vh, vs;
  int B, H, S, ntile, rows_per_split;
  int d_valid;  // channels >= d_valid are padding: not read, encoded as 0
};

// key permutation inside a 64-key tile: key -> byte position in the channel's 64-byte row.  The attention kernel
// feeds ALL 32 P values a lane holds (from the S^T accumulators: keys 32 kb2 + 8 g + 4 hh + e, kb2 < 2, g < 4, e < 4) to
// one v_mfma_scale_f32_32x32x64_f8f6f4 as the B operand, whose lane (r, hh) supplies k = 32 hh + j, j = 16 kb2 + 4 g + e
// (layout probed with exact integer data, tools/mfma_probe.hip).  Storing V^T with that k order lets the A operand be
// fetched with two 16-byte LDS reads.
__device__ __forceinline__ int vfp8_pos_of_key(int key) {
  const int kb2 = key >> 5, w = key & 31;
  const int g = w >> 3, hh = (w >> 2) & 1, e = w & 3;
  return 32 * hh + 16 * kb2 + 4 * g + e;
}

// zeroes the amax scratch (a kernel rather than hipMemsetAsync: a memset node captured into a hipGraph did not
// re-run reliably on replay - tests/test_gpu_parity.py::test_operator_is_hipgraph_capturable[fp8])
__global__ void zero_u32_kernel(unsigned* p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}

template <int DT, int D>
__global__ __launch_bounds__(256) void v_amax_kernel(VFp8Params p) {
  // channel amax over the tokens of one split (fused.cu:391-394); max is order-independent, so the
  // cross-workgroup combine is an integer atomicMax on the (non-negative) fp32 bit patterns.
  constexpr int CPR = D / 8;
  constexpr int RL = 256 / CPR;
  __shared__ float red[RL][D + 1];
  const int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int t = threadIdx.x, c = t % CPR, rl = t / CPR;
  const unsigned short* base = p.v + (int64_t)b * p.vb + (int64_t)h * p.vh + c * 8;
  const int r0 = split * p.rows_per_split, r1 = min(r0 + p.rows_per_split, p.S);
  float am[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) am[i] = 0.f;
  if (c * 8 < p.d_valid)
  for (int r = r0 + rl; r < r1; r += RL) {
    const uint4 raw = *reinterpret_cast<const uint4*>(base + (int64_t)r * p.vs);
    const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      am[2 * i] = fmaxf(am[2 * i], fabsf(load_cvt<DT>((unsigned short)(w[i] & 0xffffu))));
      am[2 * i + 1] = fmaxf(am[2 * i + 1], fabsf(load_cvt<DT>((unsigned short)(w[i] >> 16))));
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[rl][c * 8 + i] = am[i];
  __syncthreads();
  if (t < D) {
    float s = 0.f;
    for (int r = 0; r < RL; ++r) s = fmaxf(s, red[r][t]);
    atomicMax(p.amax_bits + ((int64_t)b * p.H + h) * D + t, __float_as_uint(s));
  }
}

template <int DT, int D>
__global__ __launch_bounds__(256) void v_encode_kernel(VFp8Params p) {
  // one workgroup per (tile, h, b): 64 keys x D channels -> [D][64] bytes, keys permuted
  __shared__ __attribute__((aligned(16))) uint8_t tile[D][64 + 16];
  constexpr int CPR = D / 8;
  constexpr int RPP = 256 / CPR;
  const int tl = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int t = threadIdx.x, c = t % CPR, rl = t / CPR;
  const unsigned short* base = p.v + (int64_t)b * p.vb + (int64_t)h * p.vh + c * 8;
  float inv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float amax = fmaxf(__uint_as_float(p.amax_bits[((int64_t)b * p.H + h) * D + c * 8 + i]), 1e-7f);
    inv[i] = 448.0f / amax;  // recp_scale (fused.cu:400)
    if (tl == 0 && rl == 0) p.v_scale[((int64_t)b * p.H + h) * D + c * 8 + i] = amax / 448.0f;  // fused.cu:394
  }
#pragma unroll
  for (int ps = 0; ps < 64 / RPP; ++ps) {
    const int key = ps * RPP + rl;
    const int row = tl * 64 + key;
    uint4 raw = make_uint4(0, 0, 0, 0);
    if (row < p.S && c * 8 < p.d_valid) raw = *reinterpret_cast<const uint4*>(base + (int64_t)row * p.vs);
    const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
    const int pos = vfp8_pos_of_key(key);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float v = load_cvt<DT>((unsigned short)((i & 1) ? (w[i >> 1] >> 16) : (w[i >> 1] & 0xffffu))) * inv[i];
      // v_cvt_pk_fp8_f32: OCP e4m3fn on gfx950, RNE, saturating
      const unsigned pk = __builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0u, false);
      // 16-byte chunk XOR-swizzled by the channel so the attention kernel's ds_read_b128 of 16 channels x one
      // chunk is bank-conflict-free (row stride 64 B -> 4 rows per 256-B bank window).
      const int d = c * 8 + i;
      tile[d][(((pos >> 4) ^ ((d >> 2) & 3)) << 4) | (pos & 15)] = (uint8_t)(pk & 0xffu);
    }
  }
  __syncthreads();
  uint8_t* dst = p.out + ((((int64_t)b * p.H + h) * p.ntile + tl) * D) * 64;
  // D*64 bytes, 16 B per thread per pass
  for (int idx = t; idx < D * 4; idx += 256) {
    const int d = idx >> 2, q = idx & 3;
    *reinterpret_cast<uint4*>(dst + d * 64 + q * 16) = *reinterpret_cast<const uint4*>(&tile[d][q * 16]);
  }
}

// bf16 -> fp16 of a [B, H, S, D] tensor (16 bytes per thread and step): the `v.to(float16)` of src/core.py:307-308 for the
// operators whose PV product runs on fp16 MFMAs.  HBM-bound, 4 bytes per element.
struct CastParams {
  const unsigned short* src;
  unsigned short* dst;
  int64_t sb, sh, ss, db, dh, ds;
  int B, H, S, cpr;  // cpr = 16-byte chunks per row (the valid head dim / 8)
};
__global__ __launch_bounds__(256) void cast_bf16_f16_kernel(CastParams p) {
  const int64_t n = (int64_t)p.B * p.H * p.S * p.cpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % p.cpr);
    int64_t r = i / p.cpr;
    const int s = (int)(r % p.S);
    r /= p.S;
    const int h = (int)(r % p.H), b = (int)(r / p.H);
    const uint4 raw = *reinterpret_cast<const uint4*>(p.src + b * p.sb + h * p.sh + s * p.ss + c * 8);
    unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const f16x2 pk = f16x2{(_Float16)__uint_as_float(w[e] << 16), (_Float16)__uint_as_float(w[e] & 0xffff0000u)};
      w[e] = __builtin_bit_cast(unsigned, pk);
    }
    *reinterpret_cast<uint4*>(p.dst + b * p.db + h * p.dh + s * p.ds + c * 8) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

}  // namespace lbfa

// ---------------------------------------------------------------------------------------------------
// host-side launchers (called from lbfa_api.hip)
// ---------------------------------------------------------------------------------------------------
namespace lbfa {

int mean_rows_per_split(int S) { return S <= 16384 ? 256 : 1024; }

// finalize = false: only the fp64 partial sums are produced (ws: [B,H,nsplit,D]); the caller's quantiser adds them up
// itself (QuantParams::mean_partial)
hipError_t launch_mean_seq(const void* x, int dtype, void* out, void* ws, int B, int H, int S, int D, int d_valid,
                           const int64_t* st, hipStream_t stream, bool finalize) {
  MeanParams p;
  p.d_valid = d_valid;
  p.x = (const unsigned short*)x;
  p.partial = (double*)ws;
  p.out = out;
  p.sb = st[0]; p.sh = st[1]; p.ss = st[2];
  p.B = B; p.H = H; p.S = S; p.D = D;
  p.rows_per_split = mean_rows_per_split(S);
  p.nsplit = (S + p.rows_per_split - 1) / p.rows_per_split;
  dim3 grid(p.nsplit, H, B);
#define LBFA_MEAN(DT, DD) hipLaunchKernelGGL((mean_partial_kernel<DT, DD>), grid, dim3(256), 0, stream, p)
  if (dtype == LBFA_F16) { if (D == 64) LBFA_MEAN(LBFA_F16, 64); else LBFA_MEAN(LBFA_F16, 128); }
  else { if (D == 64) LBFA_MEAN(LBFA_BF16, 64); else LBFA_MEAN(LBFA_BF16, 128); }
#undef LBFA_MEAN
  if (!finalize) return hipGetLastError();
  const int64_t n = (int64_t)B * H * D;
  dim3 g2((unsigned)((n + 255) / 256));
  if (dtype == LBFA_F16) hipLaunchKernelGGL((mean_finalize_kernel<LBFA_F16>), g2, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((mean_finalize_kernel<LBFA_BF16>), g2, dim3(256), 0, stream, p);
  return hipGetLastError();
}

#ifndef LBFA_QNB
#define LBFA_QNB 2  // 64-row blocks per workgroup of the K quantiser
#endif
hipError_t launch_quant_per_block(const QuantParams& p, int dtype, int D, int blk, hipStream_t stream) {
  const int nb = blk == 64 ? LBFA_QNB : 1;
  dim3 grid((p.nblk + nb - 1) / nb, p.H, p.B);
  const bool hm = p.mean != nullptr, hd = p.rowdot_vec != nullptr;
#define LBFA_Q(DT, DD, BB, HM, HD) hipLaunchKernelGGL((quant_per_block_kernel<DT, DD, BB, HM, HD, (BB == 64 ? LBFA_QNB : 1)>), grid, dim3(256), 0, stream, p)
#define LBFA_Q1(DT, DD, BB)                          \
  do {                                               \
    if (hm && hd) LBFA_Q(DT, DD, BB, true, true);    \
    else if (hm) LBFA_Q(DT, DD, BB, true, false);    \
    else if (hd) LBFA_Q(DT, DD, BB, false, true);    \
    else LBFA_Q(DT, DD, BB, false, false);           \
  } while (0)
#define LBFA_Q2(DT)                                         \
  do {                                                      \
    if (D == 64 && blk == 128) LBFA_Q1(DT, 64, 128);        \
    else if (D == 64 && blk == 64) LBFA_Q1(DT, 64, 64);     \
    else if (D == 128 && blk == 128) LBFA_Q1(DT, 128, 128); \
    else LBFA_Q1(DT, 128, 64);                              \
  } while (0)
  if (dtype == LBFA_F16) LBFA_Q2(LBFA_F16);
  else LBFA_Q2(LBFA_BF16);
#undef LBFA_Q2
#undef LBFA_Q1
#undef LBFA_Q
  return hipGetLastError();
}

size_t v_fp8_payload_bytes(int B, int H, int S, int D) { return (size_t)B * H * ((S + 63) / 64) * D * 64; }

hipError_t launch_quant_v_fp8(const void* v, int dtype, uint8_t* out, float* v_scale, int B, int H, int S, int D,
                              int d_valid, const int64_t* st, hipStream_t stream) {
  VFp8Params p;
  p.d_valid = d_valid;
  p.v = (const unsigned short*)v; p.out = out; p.v_scale = v_scale;
  p.amax_bits = reinterpret_cast<unsigned*>(out + v_fp8_payload_bytes(B, H, S, D));
  p.vb = st[0]; p.vh = st[1]; p.vs = st[2];
  p.B = B; p.H = H; p.S = S; p.ntile = (S + 63) / 64;
  p.rows_per_split = 512;
  const int nsplit = (S + p.rows_per_split - 1) / p.rows_per_split;
  {
    const int64_t n = (int64_t)B * H * D;
    hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p.amax_bits, n);
  }
  dim3 g1(nsplit, H, B), g2(p.ntile, H, B);
#define LBFA_V(DT, DD)                                                               \
  hipLaunchKernelGGL((v_amax_kernel<DT, DD>), g1, dim3(256), 0, stream, p);          \
  hipLaunchKernelGGL((v_encode_kernel<DT, DD>), g2, dim3(256), 0, stream, p)
  if (dtype == LBFA_F16) { if (D == 64) { LBFA_V(LBFA_F16, 64); } else { LBFA_V(LBFA_F16, 128); } }
  else { if (D == 64) { LBFA_V(LBFA_BF16, 64); } else { LBFA_V(LBFA_BF16, 128); } }
#undef LBFA_V
  return hipGetLastError();
}

hipError_t launch_cast_bf16_f16(const void* src, void* dst, int B, int H, int S, int d_valid, const int64_t* ss, const int64_t* ds,
                                hipStream_t stream) {
  CastParams p;
  p.src = (const unsigned short*)src; p.dst = (unsigned short*)dst;
  p.sb = ss[0]; p.sh = ss[1]; p.ss = ss[2];
  p.db = ds[0]; p.dh = ds[1]; p.ds = ds[2];
  p.B = B; p.H = H; p.S = S; p.cpr = d_valid / 8;
  const int64_t n = (int64_t)B * H * S * p.cpr;
  const unsigned blocks = (unsigned)((n + 255) / 256 < 256 * 64 ? (n + 255) / 256 : 256 * 64);
  hipLaunchKernelGGL(cast_bf16_f16_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, stream, p);
  return hipGetLastError();
}

// ---- per-tensor max |x| (the precision router's `compute_scale`, src/core.py:1039-1048) -----------------------------------------
// HBM-bound, 2 bytes per element read once: grid-stride over 16-byte chunks of a [B,H,S,D] view (last dim contiguous), eight
// chunks in flight per lane, the cross-workgroup combine an integer atomicMax on the (non-negative) fp32 bit patterns - max is
// order-independent, so the result is deterministic.
struct AbsmaxParams {
  const unsigned short* x;
  unsigned* out_bits;
  int64_t sb, sh, ss;
  int B, H, S, cpr;  // cpr = D / 8 chunks per row
};
template <int DT>
__global__ __launch_bounds__(256) void absmax_kernel(AbsmaxParams p) {
  const int64_t n = (int64_t)p.B * p.H * p.S * p.cpr;
  const int64_t stride = (int64_t)gridDim.x * 256;
  float am = 0.f;
  for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += 8 * stride) {
    uint4 raw[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t i = i0 + u * stride;
      raw[u] = uint4{0u, 0u, 0u, 0u};
      if (i < n) {
        const int c = (int)(i % p.cpr);
        int64_t r = i / p.cpr;
        const int sq = (int)(r % p.S);
        r /= p.S;
        const int h = (int)(r % p.H), b = (int)(r / p.H);
        raw[u] = *reinterpret_cast<const uint4*>(p.x + b * p.sb + h * p.sh + sq * p.ss + c * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned w[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // NaNs do not propagate (fmaxf drops them), as a max over |x| of finite activations expects
        am = fmaxf(am, fabsf(load_cvt<DT>((unsigned short)(w[e] & 0xffffu))));
        am = fmaxf(am, fabsf(load_cvt<DT>((unsigned short)(w[e] >> 16))));
      }
    }
  }
  am = wave_max_nonneg(am);
  if ((threadIdx.x & 63) == 0) atomicMax(p.out_bits, __float_as_uint(am));
}

hipError_t launch_absmax(const void* x, int dtype, float* out, int B, int H, int S, int D, const int64_t* st, hipStream_t stream) {
  hipLaunchKernelGGL(zero_u32_kernel, dim3(1), dim3(64), 0, stream, reinterpret_cast<unsigned*>(out), (int64_t)1);  // (not a memset node: see zero_u32_kernel)
  AbsmaxParams p;
  p.x = (const unsigned short*)x; p.out_bits = reinterpret_cast<unsigned*>(out);
  p.sb = st[0]; p.sh = st[1]; p.ss = st[2];
  p.B = B; p.H = H; p.S = S; p.cpr = D / 8;
  const int64_t n = (int64_t)B * H * S * p.cpr;
  const int64_t want = (n + 256 * 8 - 1) / (256 * 8);
  const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 256 * 16 ? 256 * 16 : want));
  if (dtype == LBFA_F16) hipLaunchKernelGGL(absmax_kernel<LBFA_F16>, dim3(blocks), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(absmax_kernel<LBFA_BF16>, dim3(blocks), dim3(256), 0, stream, p);
  return hipGetLastError();
}

}  // namespace lbfa
