// Fused low-bit FlashAttention-2 forward for gfx950 (CDNA4), FP8 PV: INT8 MFMA for QK^T, block-scaled FP8 MFMA for PV.
// (The fp16-P variants - fp16 / bf16 V, un-quantised Q / K - live in attn_fwd16.hip, on the 16x16 MFMA shapes.)
//
// Replaces the arithmetic of csrc/qattn/qk_int_sv_f8_cuda.cu:46-692 (and the `_attn_fwd` loop structure of
// src/triton/attn_qk_int8_per_block.py:24-167, ..._causal.py:24-214).  Nothing here is derived from those sources' structure; the
// tiling below is chosen for 64-wide wavefronts and the MFMA register layouts:
//
//  * one workgroup = 4 waves = one 128-row Q block (= one q_scale), each wave owns 32 query rows;
//  * K / V stream through LDS in 64-key tiles (= one k_scale each), double-buffered, by LDS-DMA (`buffer_load ... lds`)
//    issued one tile ahead.  Buffer descriptors bound every operand to its valid extent, so ragged tails read as zeros with
//    no per-lane guards, and the per-tile address update is one scalar add (voffset is loop-invariant);
//  * the score product is computed TRANSPOSED, S^T = K Q^T with v_mfma_i32_32x32x32_i8, so that a lane owns ONE query row
//    (column of S^T = lane&31) and 16 keys per 32-key block in registers: row max / row sum are in-lane reductions plus a
//    single v_permlane32_swap across the two halves;
//  * the int32 scores never pass through v_cvt_f32_i32: the MFMA accumulates on top of 1.5*2^23, whose bit pattern + s IS
//    the float 12582912+s, and the dequant scale q_scale*k_scale is folded into the exp2 argument with one v_fma;
//  * P is scaled so that its row maximum is 448 = e4m3 max (attn_utils.cuh:30), packed to e4m3 in registers - all 32 values
//    of a lane form ONE B operand - and O^T += V^T P^T takes one v_mfma_scale_f32_32x32x64_f8f6f4 per 32 channels and tile
//    (unit block scales: twice the fp16 MFMA rate); the exact row max is taken in every tile (no headroom to defer).
//
// LDS images:
//   K tile  [64 keys][D bytes]   16-B chunk c of row r stored at chunk c ^ kx(r)
//   V fp8   [D][64 bytes]        keys permuted into MFMA k order and 16-B chunks swizzled in HBM by lbfa_quant_v_fp8 -> linear copy
#include "attn_common.h"

namespace lbfa {

// (s_setprio(1) around the 2 / 4 long block-scaled MFMAs of a tile measured worse than none: -2.7 % C5, -4 % D = 64)
// Non-causal: every other ROUND of Q blocks of a head (a round = the workgroups one XCD runs at a time) walks the key tiles
// from the last one down.  The K + V panel of a head (6..8 MB at S = 16K..32K) does not fit the XCD's 4 MB L2, so every round
// streams it again; walking back, a round starts on the tiles the previous round has just left in the L2.  The direction
// depends on the Q block index only (not on what else is in the launch): results do not depend on the batch composition.
// fp8 PV: O^T += V^T P^T with ONE v_mfma_scale_f32_32x32x64_f8f6f4 per 32 channels and 64-key tile (e4m3 operands, unit
// block scales: twice the fp16 MFMA rate) - must match the V layout written by lbfa_quant_v_fp8 (quant_kernels.hip)
typedef int i32x8 __attribute__((ext_vector_type(8)));

#if defined(LBFA_STAMPS8)  // diagnostic build only (-DLBFA_STAMPS8, tools/stamps8.py): s_memtime at points of a workgroup's life, thread 0
// record per workgroup: [0..4] kernel entry, tile loop start, tile loop end, stores issued; [8..15] progress of wave 0's
// instruction stream through ONE unmasked tile (the middle one): step top, tile fetch issued, QK^T issued, row max + rescale done,
// exponentials + conversions issued, PV issued, barrier passed; [16], [17]: s_memrealtime (100 MHz) around the tile loop
__device__ long long g_stamps8[8192 * 24];
#define LBFA8_STAMP(k)                                                                                            \
  do {                                                                                                            \
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_stamps8[blockIdx.x * 24 + (k)] = __builtin_amdgcn_s_memtime();   \
  } while (0)
#define LBFA8_RSTAMP(k)                                                                                               \
  do {                                                                                                                \
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_stamps8[blockIdx.x * 24 + (k)] = __builtin_amdgcn_s_memrealtime();   \
  } while (0)
#define LBFA8_TSTAMP(k)                                     \
  do {                                                      \
    __builtin_amdgcn_sched_barrier(0);                      \
    if (ts_on) ts[k] = __builtin_amdgcn_s_memtime();        \
    __builtin_amdgcn_sched_barrier(0);                      \
  } while (0)
}  // namespace lbfa
extern "C" int lbfa_debug_stamps8(void* dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(lbfa::g_stamps8), sizeof(lbfa::g_stamps8), 0, hipMemcpyDeviceToHost);
}
namespace lbfa {
#else
#define LBFA8_STAMP(k)
#define LBFA8_RSTAMP(k)
#define LBFA8_TSTAMP(k)
#endif


// OT = dtype of O (and of the Q source when QQ): int8 Q / K codes, e4m3 V; QQ = Q is quantised inside the kernel
template <int D, int OT, bool CAUSAL, bool QQ = false>
// D = 64 fits three waves per SIMD (<= 168 registers): ask for it, or an instance one register over silently drops to two
__global__ __launch_bounds__(256, D == 64 ? 3 : 2) void attn_fwd_kernel(AttnParams p) {
  constexpr int RB = D;                              // bytes per K row
  constexpr int KS = RB / 32;                        // k-steps of the score product (32 int8 per MFMA)
  constexpr int DB = D / 32;                         // 32-channel blocks of O^T
  constexpr int KBYTES = 64 * RB;                    // K tile
  constexpr int VBYTES = 64 * D;                     // V tile ([D][64] e4m3)
  constexpr int KCH = KBYTES / (256 * 16);           // 16-B chunks per thread
  constexpr int VCH = VBYTES / (256 * 16);
  constexpr int TILES_BYTES = 2 * (KBYTES + VBYTES);
  // ONE LDS object (a second one next to an LDS-DMA target makes hipcc drain vmcnt before every ds_read): the tile
  // buffers + 16 bytes for the workgroup reduction (block amax), which must not alias a tile in flight
  __shared__ __attribute__((aligned(16))) char smem[TILES_BYTES + 16];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, hh = lane >> 5;
  LBFA8_STAMP(0);
  [[maybe_unused]] long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  [[maybe_unused]] bool ts_on = false;

  // ---- which (batch, head, q-block) -----------------------------------------------------------------
  const unsigned w_id = xcd_remap(blockIdx.x, gridDim.x);
  int qt = (int)(w_id % (unsigned)p.nQ);
  const int bh = (int)(w_id / (unsigned)p.nQ);
  if constexpr (CAUSAL) qt = p.nQ - 1 - qt;  // heaviest q-blocks of a head first
  const int b = bh / p.Hq, h = bh % p.Hq, hk = h / p.group;

  // ---- extents of this problem: dense batch entry, or sequence b of a packed variable-length batch ----
  int Sq = p.Sq, Sk = p.Sk, nK = p.nK;
  int64_t q_off = (int64_t)b * p.qb, k_off = (int64_t)b * p.kb, o_off = (int64_t)b * p.ob;
  int64_t qsc_base = (int64_t)b * p.qsc_b, ksc_base = (int64_t)b * p.ksc_b;
  if (p.cu_q != nullptr) {  // attn_qk_int8_block_varlen.py:125-141
    const int q0 = p.cu_q[b], k0 = p.cu_k[b];
    // max_seqlen_q / max_seqlen_k (p.Sq / p.Sk) size the grid and the padded scale rows: a sequence longer than the
    // caller's maximum is cut there, so that every access stays inside the buffers sized from those maxima
    Sq = min(p.cu_q[b + 1] - q0, p.Sq);
    Sk = min(p.cu_k[b + 1] - k0, p.Sk);
    if (qt * 128 >= Sq) return;  // whole workgroup, before any barrier
    nK = (Sk + 63) >> 6;
    q_off = (int64_t)q0 * p.qs;
    k_off = (int64_t)k0 * p.ks;
    o_off = (int64_t)q0 * p.os;
    if (p.cu_qscale != nullptr) {
      qsc_base = (int64_t)p.cu_qscale[b] * p.qsc_b;
      ksc_base = (int64_t)p.cu_kscale[b] * p.ksc_b;
    }
  }

  const int row0 = qt * 128 + wave * 32;  // first query row of this wave
  const int qrow = row0 + r;

  // ---- operand windows (bytes).  The descriptor is re-based per tile with scalar arithmetic, so the
  // hardware range check sees only the loop-invariant per-lane offset.
  const char* kbase = (const char*)p.k + (k_off + (int64_t)hk * p.kh);
  const int64_t k_bytes = (int64_t)(Sk - 1) * p.ks + D;
  const int64_t k_tile_stride = 64 * p.ks;
  const char* vbase = (const char*)p.v + (((int64_t)b * p.Hkv + hk) * nK) * (int64_t)(D * 64);
  const int64_t v_bytes = (int64_t)nK * D * 64;
  const int64_t v_tile_stride = D * 64;  // bytes between consecutive 64-key tiles

  // ---- loop-invariant per-thread offsets of the tile fetch ------------------------------------------------------
  // Chunk i of a thread is chunk 0 moved down by a whole number of rows (KROWS per pass), which leaves the swizzle
  // unchanged: one voffset per operand, the rest is a scalar soffset.
  constexpr int KCPR = RB / 16, KROWS = 256 / KCPR;  // K: 16-B chunks per row, rows per pass
  unsigned k_goff;
  {
    const int row = t / KCPR, ch = t % KCPR;
    // LDS-DMA writes a wave's 64 x 16 bytes linearly (thread t -> byte 16 t of the pass): the swizzle moves to the SOURCE
    // address - the slot (row, ch) of the image holds global chunk ch ^ kx(row)
    k_goff = (unsigned)row * (unsigned)p.ks + ((ch ^ kx<RB>(row)) << 4);
  }
  const unsigned v_goff = t * 16;  // the V image is copied as it lies in HBM
  const unsigned k_gstep = KROWS * (unsigned)p.ks;  // bytes between a thread's K chunks
  static_assert(KROWS * RB == 4096, "one pass of 256 threads x 16 bytes");
  // windows are < 2 GiB (checked by the C ABI): remaining bytes in 32-bit scalar arithmetic
  const int k_bytes32 = (int)k_bytes, v_bytes32 = (int)v_bytes, k_stride32 = (int)k_tile_stride, v_stride32 = (int)v_tile_stride;
  typedef __attribute__((address_space(3))) void* lds_void_ptr;
  // fetch tile j into LDS buffer `buf_tag`; rows / tiles past the end are outside the descriptor and read as zeros
  auto load_tile = [&](int j, auto buf_tag) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_tag)::value;
    const bool in_range = (unsigned)j < (unsigned)nK;  // the look-ahead past either end gets a window of 0 bytes
    const int ko = in_range ? j * k_stride32 : 0, vo = in_range ? j * v_stride32 : 0;
    const int k_rem = in_range ? max(0, k_bytes32 - ko) : 0, v_rem = in_range ? max(0, v_bytes32 - vo) : 0;
    const __amdgpu_buffer_rsrc_t k_rs = make_rsrc(kbase + ko, (unsigned)k_rem);
    const __amdgpu_buffer_rsrc_t v_rs = make_rsrc(vbase + vo, (unsigned)v_rem);
    // DMA destination = wave-uniform base (+ 16 bytes per lane, implicit)
    char* kdst = smem + BUF * KBYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < KCH; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rs, (lds_void_ptr)(kdst + i * 4096), 16, (int)k_goff, (int)(i * k_gstep), 0, 0);
    char* vdst = smem + 2 * KBYTES + BUF * VBYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < VCH; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rs, (lds_void_ptr)(vdst + i * 4096), 16, (int)v_goff, (int)(i * 4096), 0, 0);
  };

  // processing order of the key tiles: i-th tile processed = tile_of(i)
  constexpr int kRound = 64;
  const bool rev = !CAUSAL && ((Sk & 63) == 0) && (((qt / kRound) & 1) != 0);  // workgroup-uniform
  auto tile_of = [&](int i) __attribute__((always_inline)) { return rev ? nK - 1 - i : i; };
  load_tile(tile_of(0), std::integral_constant<int, 0>{});  // first K / V tile: in flight while Q is fetched (and quantised)
  // ... and so are the dequantisation scales of the first 64 key tiles (lane l: tile l), needed right after the Q prologue
  const float* ksc = p.k_scale + ksc_base + (int64_t)hk * p.ksc_h;
  const int ksc_blk = (int)p.ksc_blk;
  const float ks_first = lane < nK ? ksc[lane * ksc_blk] : 0.f;

  // ---- Q fragments (B operand of the score MFMA): lane (r, hh) holds bytes [32s+16hh, +16) of its row.
  // Rows >= Sq are out of the descriptor's range and read as zeros.
  i32x4 qf[KS];
  float qsc = 1.0f;
  float row_corr = 0.f;  // QQ: lse correction q . km of this lane's row
  if constexpr (QQ) {
    // Quantise this workgroup's 128 x D block of Q here instead of in a pre-pass (src/triton/quant_per_block.py:132-178,
    // same arithmetic as quant_per_block_kernel: x * sm_scale -> block amax -> scale = amax / qmax -> RN(x / scale) ->
    // round half away): the lane needs exactly the 16 * KS elements it feeds to the MFMA.
    const char* qsrc = (const char*)p.q + 2 * (q_off + (int64_t)h * p.qh);
    const __amdgpu_buffer_rsrc_t q_rs = make_rsrc(qsrc, (unsigned)(2 * ((int64_t)(Sq - 1) * p.qs + p.d_valid)));
    float xs[KS][16];
    float amax = 0.f, dot = 0.f;
    const unsigned short* vec = p.q_dot_vec ? p.q_dot_vec + ((int64_t)b * p.Hkv + hk) * D : nullptr;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int col = 32 * s + 16 * hh + 8 * hf;
        const u32x4 raw = buf_load16(q_rs, col < p.d_valid ? 2 * ((unsigned)qrow * (unsigned)p.qs + col) : 0x80000000u, 0);
        float xv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xv[e] = load_cvt<OT>((unsigned short)((e & 1) ? (raw[e >> 1] >> 16) : (raw[e >> 1] & 0xffffu)));
          const float x = xv[e] * p.q_sm_scale;
          xs[s][8 * hf + e] = x;
          amax = fmaxf(amax, fabsf(x));
        }
        if (vec != nullptr) {  // wave-uniform: only when the caller wants the LSE with smooth-K
          const u32x4 vraw = *reinterpret_cast<const u32x4*>(vec + col);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            dot += xv[e] * load_cvt<OT>((unsigned short)((e & 1) ? (vraw[e >> 1] >> 16) : (vraw[e >> 1] & 0xffffu)));
        }
      }
    row_corr = load_cvt<OT>(store_cvt<OT>(half_swap_sum(dot)));  // rounded to the storage dtype (src/core.py:294-304)
    amax = wave_max_nonneg(amax);
    float* red = reinterpret_cast<float*>(smem + TILES_BYTES);
    if (lane == 0) red[wave] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();  // red[] is read by everyone before the overflow vote may reuse it
    const float scale = fmaxf(amax, 1e-7f) / p.q_qmax;
    qsc = scale;
    const float rcp = 1.0f / scale;
    const bool exact_rcp_ok = (__builtin_amdgcn_readfirstlane(__float_as_uint(scale)) & 0x7fffffu) != 0x7fffffu;
    auto encode = [&](auto fast_tag) __attribute__((always_inline)) {  // block-uniform choice, two straight-line instances (as in quant_kernels.hip)
      constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        unsigned w[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          int qv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xv = xs[s][4 * g4 + e];
            float y;
            if constexpr (FAST) {  // Markstein: RN(x / scale) from the correctly rounded reciprocal (see quant_kernels.hip)
              const float q0 = xv * rcp;
              y = __builtin_fmaf(__builtin_fmaf(-q0, scale, xv), rcp, q0);
            } else {
              y = xv / scale;
            }
            qv[e] = (int)(y + __builtin_copysignf(0.5f, y));
          }
          const unsigned p01 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pk_i16(qv[0], qv[1]));
          const unsigned p23 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pk_i16(qv[2], qv[3]));
          w[g4] = __builtin_amdgcn_perm(p23, p01, 0x06040200u);
        }
        qf[s] = i32x4{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
      }
    };
    if (exact_rcp_ok) encode(std::true_type{});
    else encode(std::false_type{});
  } else {
    const char* qbase = (const char*)p.q + (q_off + (int64_t)h * p.qh);
    const __amdgpu_buffer_rsrc_t q_rs = make_rsrc(qbase, (unsigned)((int64_t)(Sq - 1) * p.qs + D));
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const unsigned col_b = 16 * hh + 32 * s;  // byte column
      u32x4 raw = buf_load16(q_rs, (unsigned)qrow * (unsigned)p.qs + col_b, 0);
      qf[s] = __builtin_bit_cast(i32x4, raw);
    }
    qsc = p.q_scale[qsc_base + (int64_t)h * p.qsc_h + (int64_t)qt * p.qsc_blk];
  }

  int n_tiles = nK;
  if constexpr (CAUSAL) n_tiles = min(nK, 2 * (qt + 1));

  // Fragment read addresses.  The swizzles depend only on the low row bits, so the block / k-step / buffer parts are
  // compile-time byte offsets folded into the ds_read immediates.
  // K: chunk (2s + hh) ^ kx(r) of row r = kf_lane ^ (s << 5) with kf_lane = r * RB + ((hh ^ kx(r)) << 4)
  const unsigned kf_lane = r * RB + ((hh ^ kx<RB>(r)) << 4);
  unsigned kf_base[KS];  // + kb2 * 32 * RB
#pragma unroll
  for (int s = 0; s < KS; ++s) kf_base[s] = kf_lane ^ (s << 5);
  // V: channel row r (+ 32 db) of the [D][64] image, this lane's 32 keys = 16-B chunks 2hh and 2hh+1 (the second at ^ 16),
  // chunk c stored at c ^ ((r>>2)&3)
  const unsigned vf_base = 2 * KBYTES + r * 64 + (((2 * hh) ^ ((r >> 2) & 3)) << 4);

  // ---- running state --------------------------------------------------------------------------------
  // Row sums: fp32 adds of P BEFORE it is rounded to e4m3, as the reference sums (qk_int_sv_f8_cuda.cu:430-445).  Each lane
  // sums the keys IT has seen (its half of every tile); both halves of a row share m_run, hence every rescale factor, so the
  // halves are added once, in the epilogue.
  f32x16 acc_o[DB];
  float m_run = -INFINITY;  // running row max (base-2 domain), identical in both halves of a row
  float m_lo = 0.f;         // `wide` waves: the row max is the exact product (integer score) x (scale) = m_run + m_lo (see below)
  float l_run = 0.f;
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc_o[db][i] = 0.f;

  // The int8 MFMA accumulates on top of this constant block (kept in registers for the whole kernel).
  i32x16 cmagic;
#pragma unroll
  for (int i = 0; i < 16; ++i) cmagic[i] = kMagicBits;
  asm volatile("" : "+v"(cmagic));  // opaque: otherwise the block is re-materialised from SGPRs with 8 v_mov_b64 in every tile

  // ---- bias folding --------------------------------------------------------------------------------
  // With tv = kMagic + s (the accumulator bits) the exponent argument is ONE fma:
  //     s*sc - m  ==  fma(tv, sc, c1),   c1 = -kMagic*sc - m + kFp8Offset
  // The per-tile scale is rounded to a multiple of g = G/2^22, G a power of two chosen from the largest dequantisation scale
  // of this (batch, kv-head) (relative change <= 2^-21 * sc_max/sc, far below int8 quantisation noise), so kMagic*sc is exact;
  // m is the exact row max (P_max = 448 = e4m3 max exactly, attn_utils.cuh:30); c1 is rounded to fp32 twice, by <= ulp(kMagic sc)
  // ~ 0.3 G in all - invisible at 3 mantissa bits while G is small.
  // WIDE scores (round 4).  That rounding, and the 2^-19-relative rounding of the scale, stop being invisible when the scores are
  // hundreds of binades wide - the reference's bench distribution randint(-100, 100): G = 1..2, references ~ +-5000 - : tiles drift
  // against each other by up to 0.24 G binades, and at D = 128 (|kMagic sc| ~ 1e6, ulp 2^-4) c1 alone could lift the largest P past
  // 464, the last value that still rounds to 448: v_cvt_pk_fp8_f32 does not saturate, it returns the NaN code, and the query's
  // whole output row became NaN (found by tools/soak.py).  A wave therefore leaves the grid - un-rounded scale q_scale k_scale[j],
  // bias taken off the scores with one exact subtraction each, c1 = 8.807 - m - from the start when G >= 2^-6 (kWideG), and from
  // the tile in which one of its row maxima leaves +-2^7 binades (kGridRef) otherwise; 32 VALU more per tile where it applies.
  float ks_max = ks_first;
  for (int i = lane + 64; i < nK; i += 64) ks_max = fmaxf(ks_max, ksc[i * ksc_blk]);
  ks_max = fmaxf(wave_max_nonneg(ks_max), 1e-30f);  // scales are positive (the quantiser floors amax)
  const float sc_max = qsc * ks_max;
  const int gexp = __builtin_amdgcn_readfirstlane((int)((__float_as_uint(1.25f * kMagic * sc_max) >> 23) & 0xff) - 127 + 1 - 21);  // log2(G)
  const float g = __builtin_ldexpf(1.0f, gexp - 22), invg = __builtin_ldexpf(1.0f, 22 - gexp);
  constexpr int kWideG = -6;            // log2 of the grid step from which every tile dequantises un-rounded
  constexpr float kGridRef = 128.0f;    // |row max| (binades) beyond which a wave leaves the grid (as attn_fwd16.hip)
  bool wide = gexp >= kWideG;           // wave-uniform; only ever switched on
  constexpr float kHugeRef = 0x1p16f;   // |row max| (binades) beyond which a `wide` wave re-references its scores (compute_tile)
  bool huge = false;                    // wave-uniform; only ever switched on
  // Per-tile constants sc (dequantisation scale on the g grid) and c0 = -kMagic * sc are the same for every lane: lane l
  // of the wave computes them for tile 64 c + l once per chunk of 64 tiles, and each tile fetches its pair with two
  // v_readlane (no per-tile global load, no per-tile float math on uniform values).
  // (sc_tab, c0_tab: the scale and the bias constant as the tile loop uses them - on the grid, or un-rounded and 0 for a `wide`
  // wave; ks_tab: the raw k_scale, from which a wave that leaves the grid in the middle of its tiles rebuilds them without a memory
  // access)
  float sc_tab = 0.f, c0_tab = 0.f, ks_tab = 0.f;
  auto refresh_scale_table = [&](int j0) __attribute__((always_inline)) {
    const int jt = j0 + lane;
    const float ks_l = j0 == 0 ? ks_first : (jt < nK ? ksc[jt * ksc_blk] : 0.f);
    ks_tab = ks_l;
    // at least one grid step: a block whose scale is < 2^-22 of the largest one (an all-zero K block) must not get
    // sc = 0, or a masked key (tv = -inf) would turn into fma(-inf, 0, c1) = NaN
    sc_tab = fmaxf(__builtin_rintf(qsc * ks_l * invg), 1.0f) * g;
    c0_tab = -kMagic * sc_tab;  // exact
    if (wide) {
      sc_tab = fmaxf(qsc * ks_l, 1e-30f);
      c0_tab = 0.f;
    }
  };

  // One 64-key tile.
  auto compute_tile = [&](auto buf_tag, int j, auto masked_tag) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_tag)::value;
    constexpr bool MASKED = decltype(masked_tag)::value;
    const char* kbuf = smem + BUF * KBYTES;
    const char* vbuf = smem + BUF * VBYTES;
    float sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc_tab), j & 63));
    float c0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c0_tab), j & 63));
    const float bias = wide ? kMagic : 0.f;  // wave-uniform (scalar select)
    float x[2][16];  // scores as floats kMagic + s (accumulator bits), overwritten in place by P
    // -- S^T = K Q^T (int8 -> int32, biased by kMagic): one 32-key block
    auto compute_scores = [&](auto kb2_tag) __attribute__((always_inline)) {
      constexpr int kb2 = decltype(kb2_tag)::value;
      i32x16 sacc;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const i32x4 kf = *reinterpret_cast<const i32x4*>(kbuf + kf_base[s] + kb2 * 32 * RB);
        if (s == 0) sacc = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf, qf[s], cmagic, 0, 0, 0);
        else sacc = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf, qf[s], sacc, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float tv = __int_as_float(sacc[i]);
        if constexpr (MASKED) {
          const int key = j * 64 + 32 * kb2 + (i & 3) + 8 * (i >> 2) + 4 * hh;
          bool dead = key >= Sk;
          if constexpr (CAUSAL) dead = dead || (key > qrow);
          if (dead) tv = -INFINITY;  // fma(-inf, sc, c1) = -inf -> p = 0  (sc > 0, see refresh_scale_table)
        }
        x[kb2][i] = tv;
      }
    };
    compute_scores(std::integral_constant<int, 0>{});
    compute_scores(std::integral_constant<int, 1>{});
    LBFA8_TSTAMP(2);
    // -- online softmax, base 2: move m_run up to this tile's row max, rescaling O and l, when some row of the wave needs it
    // (first tile: m_run = -inf -> alpha = 0)
    {
      // the lane's 32 scores as ORDER KEYS (attn_common.h: the accumulator bits of kMagic + s compare like s as signed integers,
      // -inf sorts below all of them): 15 v_max3_i32 + 1 v_max_i32, no canonicalising v_max in front as an fmaxf on MFMA output
      // gets, then the other half of the row through one v_permlane32_swap
      float tmax = key_max3<true>(x[0][0], x[0][1], x[0][2]);
#pragma unroll
      for (int i = 3; i < 15; i += 2) tmax = key_max3<true>(tmax, x[0][i], x[0][i + 1]);
      tmax = key_max3<true>(tmax, x[0][15], x[1][0]);
#pragma unroll
      for (int i = 1; i < 15; i += 2) tmax = key_max3<true>(tmax, x[1][i], x[1][i + 1]);
      tmax = key_max<true>(tmax, x[1][15]);
      {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
        tmax = key_max<true>(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      }
      // row max of the dequantised scores (-inf if all masked); bias, c0 = 0, -kMagic sc on the grid, kMagic, 0 for a `wide` wave
      float xmax = __builtin_fmaf(tmax - bias, sc, c0);
      float m_cand = fmaxf(m_run, xmax);
      if (__any(m_cand > m_run)) {
        if (!wide) {  // wave-uniform: does the new row maximum leave the range in which the grid is exact enough?
          const float ra = __builtin_fabsf(m_cand);
          if (__any(ra > kGridRef && ra < INFINITY)) {
            wide = true;  // this tile and every later one of the wave (the earlier ones had all their maxima inside the range)
            sc_tab = fmaxf(qsc * ks_tab, 1e-30f);
            c0_tab = 0.f;
            sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc_tab), j & 63));
            c0 = 0.f;
            xmax = (tmax - kMagic) * sc;  // exact integer times the un-rounded scale
            m_cand = fmaxf(m_run, xmax);
          }
        }
        float dlo = 0.f;
        if (wide) {
          // (one compare: a row that has not seen a key yet, m_cand = -inf, switches the wave too - that costs time, nothing else)
          if (!huge) huge = __any(__builtin_fabsf(m_cand) > kHugeRef) != 0;
          if (huge) {  // the rows that take this tile's maximum keep what the product lost when it was rounded to m_run
            // (references taken before the switch keep m_lo = 0: they were below kHugeRef, their remainder below 2^-9 binades)
            const float lo = __builtin_fmaf(tmax - kMagic, sc, -xmax);  // exact (NaN for a row without keys: never selected)
            const float lo_new = xmax > m_run ? lo : m_lo;
            dlo = m_lo - lo_new;
            m_lo = lo_new;
          }
        }
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_cand) + dlo);  // m_run = -inf -> 0
        m_run = m_cand;
        l_run *= alpha;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc_o[db][i] *= alpha;
      }
    }
    LBFA8_TSTAMP(3);
    float c1 = c0 - m_run + kFp8Offset;  // +inf while m_run = -inf
    if (wide) {  // wave-uniform: scores as the integers themselves (-inf stays -inf)
      float bias_r = kMagic;
      if (huge) {
        // ... as integers RELATIVE TO THE ROW'S REFERENCE, in this tile's units.  The reference of a row is an exact product
        // t x sc of some earlier tile, kept as m_run + m_lo; with r = the integer nearest to it in this tile's units,
        //     s sc - m = (s - r) sc + (r sc - m_run - m_lo):
        // the first product is small where P is not, the bracket is the remainder of one fma.  One fma against the ROUNDED m_run
        // leaves P_max = 448 x 2^(rounding of m_run): beyond |m_run| ~ 2^19 binades (q, k some hundred times N(0,1)) that is
        // percents of P, saturated away by the conversion while the row sum keeps them; below kHugeRef it is < 0.3 % and the plain
        // form stays.  |r| <= 2^22 keeps kMagic + r exact; a clamped r only gives up the precision, where P = 0 anyway.
        const float r = __builtin_amdgcn_fmed3f(__builtin_rintf(m_run * __builtin_amdgcn_rcpf(sc)), -0x1p22f, 0x1p22f);
        bias_r = kMagic + r;
        c1 = (__builtin_fmaf(r, sc, -m_run) - m_lo) + kFp8Offset;
      }
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
        for (int i = 0; i < 16; ++i) x[kb2][i] -= bias_r;
    }
    // V^T fragment of channel block 0: requested here, in flight under the exponentials
    i32x4 vnext[2][2];
    vnext[0][0] = *reinterpret_cast<const i32x4*>(vbuf + vf_base);
    vnext[0][1] = *reinterpret_cast<const i32x4*>(vbuf + (vf_base ^ 16u));
    __builtin_amdgcn_sched_barrier(0);
    // -- P, its row sums, and the packed P^T operand: all 32 P values of the lane (k = 32 hh + 16 kb2 + i)
    i32x8 pf8;
    float psum = 0.f;
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        x[kb2][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[kb2][i], sc, c1));
        psum += x[kb2][i];
        // keep P in the score registers: left alone, the scheduler issues all exponentials first and sinks the adds, which
        // needs 32 more registers
        if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int g8 = 0; g8 < 2; ++g8) {
        const int ks = 2 * kb2 + g8, rb = 8 * g8;
        unsigned w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(x[kb2][rb + 0], x[kb2][rb + 1], w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(x[kb2][rb + 2], x[kb2][rb + 3], w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(x[kb2][rb + 4], x[kb2][rb + 5], w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(x[kb2][rb + 6], x[kb2][rb + 7], w1, true);
        pf8[2 * ks] = (int)w0;
        pf8[2 * ks + 1] = (int)w1;
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // phase fence: keeps the V fragment reads and the row-sum adds where they are (registers)
    l_run += psum;
    LBFA8_TSTAMP(4);
    // -- O^T += V^T P^T: e4m3 x e4m3 (cbsz = blgp = 0), E8M0 block scales 0x7F = 2^0.  The V^T fragment of channel block db + 1 is
    // requested before the MFMA of block db is issued (the first one before the exponentials, see above): a 16-pass MFMA covers the
    // LDS round trip of the next operand instead of waiting for its own.
    static_for<0, DB>([&](auto i) {
      constexpr int db = decltype(i)::value;
      if constexpr (db + 1 < DB) {
        vnext[(db + 1) & 1][0] = *reinterpret_cast<const i32x4*>(vbuf + vf_base + (db + 1) * 2048);
        vnext[(db + 1) & 1][1] = *reinterpret_cast<const i32x4*>(vbuf + (vf_base ^ 16u) + (db + 1) * 2048);
      }
      const i32x4 v0 = vnext[db & 1][0], v1 = vnext[db & 1][1];
      const i32x8 vf = i32x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      acc_o[db] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, pf8, acc_o[db], 0, 0, 0, 0x7F, 0, 0x7F);
      if constexpr (db + 1 < DB) __builtin_amdgcn_sched_barrier(0);  // keeps the next block's reads in front of this MFMA
    });
    // Toolchain work-around (ROCm 7.2 / clang 22): the wait states the compiler leaves between this 16-pass MFMA and
    // a VALU read of its result (register copies at control-flow edges, the epilogue) are too few - the last two
    // accumulator registers were read stale (tests: odd tile counts).  LBFA_MX_NOP more wait states close the gap
    // (checked on the shipped code object by tools/check_mfma_hazards.py).
#ifndef LBFA_MX_NOP
#define LBFA_MX_NOP 7
#endif
#if LBFA_MX_NOP >= 0  // (-1: build without the pad, for tools/check_mfma_hazards.py to show what the compiler leaves)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop %0" ::"n"(LBFA_MX_NOP));
    __builtin_amdgcn_sched_barrier(0);
#endif
  };

  // ---- tile loop: one barrier per tile, buffers alternate statically (loop unrolled by two).
  // Full (unmasked) tiles first, branch-free; then the at most three tiles that need masking
  // (causal diagonal block = 2 tiles, ragged last tile).
  int n_main = n_tiles;
  if constexpr (CAUSAL) n_main = min(n_tiles, 2 * qt);
  else if ((Sk & 63) != 0) n_main = n_tiles - 1;

  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  using No = std::false_type;
  using Yes = std::true_type;
  auto step = [&](auto buf_tag, auto nbuf_tag, int i, auto masked_tag) __attribute__((always_inline)) {
    const int j = tile_of(i);
#if defined(LBFA_STAMPS8)
    ts_on = !decltype(masked_tag)::value && i == (n_main >> 1) && wave == 0;
#endif
    LBFA8_TSTAMP(0);
    if (i != 0 && (j & 63) == (rev ? 63 : 0)) refresh_scale_table(j & ~63);  // wave-uniform: entering the next chunk of 64 tiles
    load_tile(tile_of(i + 1), nbuf_tag);
    LBFA8_TSTAMP(1);
    bool skip = false;
    if constexpr (decltype(masked_tag)::value && CAUSAL) skip = j * 64 > row0 + 31;  // all keys above all rows of this wave
    if (!skip) compute_tile(buf_tag, j, masked_tag);
    LBFA8_TSTAMP(5);
    __syncthreads();  // with LDS-DMA in flight this waits vmcnt(0) first: tile j + 1 has landed when the barrier opens
    LBFA8_TSTAMP(6);
  };

  refresh_scale_table(tile_of(0) & ~63);
  __syncthreads();
  // P -> e4m3 must SATURATE (the reference converts with cvt.rn.satfinite.e4m3x2.f32, csrc/numeric_conversion.cuh:39-54): with the
  // wave's MODE.FP16_OVFL clear, v_cvt_pk_fp8_f32 turns everything beyond 464 into the NaN code 0x7f; with it set, finite values
  // clamp to +-448 (probed on gfx950, tools/fp8_sat_probe.hip).  P stays <= 448 by construction - this is the belt to those braces.
  // The bit is set for the TILE LOOP only: it also makes fp32 -> fp16 conversions clamp instead of overflowing to inf, and the
  // prologue's q . km (rounded to the storage dtype, src/core.py:294-304) and the epilogue's stores overflow as the reference's do.
  __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);  // hwreg(HW_REG_MODE, offset 23, 1 bit) = FP16_OVFL
  LBFA8_STAMP(1);
  LBFA8_RSTAMP(16);
  {
    int j = 0;
    for (; j + 1 < n_main; j += 2) {
      step(B0{}, B1{}, j, No{});
      step(B1{}, B0{}, j + 1, No{});
    }
    // here j is even: tile j lives in buffer 0
    for (; j < n_tiles; j += 2) {
      if (j < n_main) step(B0{}, B1{}, j, No{});
      else step(B0{}, B1{}, j, Yes{});
      if (j + 1 < n_tiles) {
        if (j + 1 < n_main) step(B1{}, B0{}, j + 1, No{});
        else step(B1{}, B0{}, j + 1, Yes{});
      }
    }
  }
  LBFA8_STAMP(2);
  LBFA8_RSTAMP(17);
  __builtin_amdgcn_s_setreg(1 | (23 << 6), 0);  // fp16 overflow semantics back to the default for the epilogue
  const float l_tot = half_swap_sum(l_run);

  // ---- epilogue: O = O^T / l x v_scale, LSE ------------------------------------------------------------
  const float inv_l = l_tot > 0.f ? 1.0f / l_tot : 0.f;  // a sequence without keys (packed batches only) yields zeros
  if (qrow < Sq) {
    unsigned short* op = reinterpret_cast<unsigned short*>(p.o) + o_off + (int64_t)h * p.oh + (int64_t)qrow * p.os;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int d0 = 32 * db + 8 * g4 + 4 * hh;
        if (d0 >= p.d_valid) continue;  // d_valid is a multiple of 8
        const f32x4 vs4 = *reinterpret_cast<const f32x4*>(p.v_scale + ((int64_t)b * p.Hkv + hk) * D + d0);
        float o4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = acc_o[db][4 * g4 + e] * inv_l * vs4[e];
        uint2 pk;
        pk.x = (unsigned)store_cvt<OT>(o4[0]) | ((unsigned)store_cvt<OT>(o4[1]) << 16);
        pk.y = (unsigned)store_cvt<OT>(o4[2]) | ((unsigned)store_cvt<OT>(o4[3]) << 16);
        *reinterpret_cast<uint2*>(op + d0) = pk;
      }
    if (p.lse != nullptr && hh == 0) {
      float ls = log2f(l_tot) + m_run - kFp8Offset;  // base-2 domain (attn_qk_int8_per_block.py:164-167, qk_int_sv_f8_cuda.cu:689)
      const int64_t li = ((int64_t)b * p.Hq + h) * p.Sq + qrow;  // dense only (the packed entry points take no lse)
      ls *= p.lse_scale;
      if constexpr (QQ) ls += row_corr * p.lse_corr_scale;  // 0 when there is no smoothing vector
      else if (p.lse_corr != nullptr) ls += p.lse_corr[li] * p.lse_corr_scale;
      p.lse[li] = ls;
    }
  }
  LBFA8_STAMP(3);
#if defined(LBFA_STAMPS8)
  if (threadIdx.x == 0 && blockIdx.x < 8192)
    for (int k = 0; k < 8; ++k) g_stamps8[blockIdx.x * 24 + 8 + k] = ts[k];
#endif
}

// ---- launchers: every fp16-P variant runs in attn_fwd16.hip, fp8 PV here ------------------------------------------
hipError_t launch16_attn_fwd_d64(const AttnParams& p, int v_dtype, int o_dtype, int causal, hipStream_t stream);
hipError_t launch16_attn_fwd_d128(const AttnParams& p, int v_dtype, int o_dtype, int causal, hipStream_t stream);
hipError_t launch16_attn_fwd_qq_d64(const AttnParams& p, int dtype, int v_dtype, int causal, hipStream_t stream);
hipError_t launch16_attn_fwd_qq_d128(const AttnParams& p, int dtype, int v_dtype, int causal, hipStream_t stream);
hipError_t launch16_attn_fwd_f16_d64(const AttnParams& p, int dtype, int causal, hipStream_t stream);
hipError_t launch16_attn_fwd_f16_d128(const AttnParams& p, int dtype, int causal, hipStream_t stream);

#define LBFA_LAUNCH_FP8(DD, OT, QQ)                                                                    \
  do {                                                                                                 \
    if (causal) hipLaunchKernelGGL((attn_fwd_kernel<DD, OT, true, QQ>), grid, block, 0, stream, p);    \
    else hipLaunchKernelGGL((attn_fwd_kernel<DD, OT, false, QQ>), grid, block, 0, stream, p);          \
  } while (0)

// int8 Q / K codes; V fp16 / e4m3 ([D][64]-per-tile image of lbfa_quant_v_fp8)
hipError_t launch_attn_fwd(const AttnParams& p, int D, int v_dtype, int o_dtype, int causal, hipStream_t stream) {
  if (v_dtype != LBFA_E4M3)
    return D == 64 ? launch16_attn_fwd_d64(p, v_dtype, o_dtype, causal, stream) : launch16_attn_fwd_d128(p, v_dtype, o_dtype, causal, stream);
  dim3 grid((unsigned)p.B * p.Hq * p.nQ), block(256);
  if (D == 64) { if (o_dtype == LBFA_F16) LBFA_LAUNCH_FP8(64, LBFA_F16, false); else LBFA_LAUNCH_FP8(64, LBFA_BF16, false); }
  else { if (o_dtype == LBFA_F16) LBFA_LAUNCH_FP8(128, LBFA_F16, false); else LBFA_LAUNCH_FP8(128, LBFA_BF16, false); }
  return hipGetLastError();
}

// int8 K codes, Q quantised inside the kernel from its fp16 / bf16 source (dtype = Q's = O's); V fp16 or e4m3
hipError_t launch_attn_fwd_qq(const AttnParams& p, int D, int dtype, int v_dtype, int causal, hipStream_t stream) {
  if (v_dtype != LBFA_E4M3)
    return D == 64 ? launch16_attn_fwd_qq_d64(p, dtype, v_dtype, causal, stream) : launch16_attn_fwd_qq_d128(p, dtype, v_dtype, causal, stream);
  dim3 grid((unsigned)p.B * p.Hq * p.nQ), block(256);
  if (D == 64) { if (dtype == LBFA_F16) LBFA_LAUNCH_FP8(64, LBFA_F16, true); else LBFA_LAUNCH_FP8(64, LBFA_BF16, true); }
  else { if (dtype == LBFA_F16) LBFA_LAUNCH_FP8(128, LBFA_F16, true); else LBFA_LAUNCH_FP8(128, LBFA_BF16, true); }
  return hipGetLastError();
}
#undef LBFA_LAUNCH_FP8

// un-quantised Q / K / V of one dtype
hipError_t launch_attn_fwd_f16(const AttnParams& p, int D, int dtype, int causal, hipStream_t stream) {
  return D == 64 ? launch16_attn_fwd_f16_d64(p, dtype, causal, stream) : launch16_attn_fwd_f16_d128(p, dtype, causal, stream);
}

}  // namespace lbfa
