// Fused low-bit FlashAttention-2 forward for gfx950, fp16-P variants, on the 16x16 MFMA shapes
// (v_mfma_i32_16x16x64_i8 for QK^T, v_mfma_f32_16x16x32_f16 for PV).
//
// Same operator and the same tiling in the large as the fp8 kernel (attn_fwd.hip): one workgroup = 4 waves = one 128-row Q block,
// 32 query rows per wave, 64-key K / V tiles by LDS-DMA, double-buffered, scores biased by 1.5 * 2^23 so the int32 accumulator
// bits are floats, dequantisation folded into the exp2 argument - but every matrix product is cut into 16x16 output blocks.  Why: on this part the same FLOPs cost fewer cycles AND hold a higher clock on the
// 16x16 shapes (tools/ubench_tile.hip: the MFMA + softmax stream of one tile without memory traffic takes 290 ns per wave
// and SIMD on them against 400 ns on the 32x32 shapes: -14 % cycles, +19 % clock; MI355X_MICROARCH.md "DVFS give-back" item 7).
// At D = 64 one i8 MFMA spans the whole head dim (K = 64): the eight score MFMAs of a tile are independent - no accumulate
// chains in front of the softmax - and the first blocks are ready while the later ones still run.
//
// Lane roles (i = lane & 15, g = lane >> 4):
//   S^T = K Q^T   A = K fragment: key 16 kb + i, row bytes [64 s + 16 g, +16)      (ds_read_b128, one per (kb, s), both row blocks)
//                 B = Q fragment: query 16 rb + i of the wave, same bytes            (registers, whole kernel)
//                 C: lane holds, for ITS query 16 rb + i, the keys 16 kb + 4 g + {0..3}: 2 x 4 x 4 = 32 scores per tile
//   O^T += V^T P^T  B = P^T fragment of k-step s (32 keys): element j = score of key 32 s + 16 (j >> 2) + 4 g + (j & 3) - the
//                 lane's own registers of key blocks 2 s and 2 s + 1, converted pairwise: P never touches LDS
//                 A = V^T fragment with the SAME key order: two ds_read_b64_tr_b16 (rows 32 s + 4 g + {0..3} and + 16) of the
//                 row-major V tile; one fragment serves both row blocks
//                 C: lane holds, for its query, channels 16 cb + 4 g + {0..3}
//   A query row's scores live in four lanes (g = 0..3): a row max needs two cross-group steps, which only the exact path (masked
//   tiles, re-run) takes; the row sums come out of one more PV block against an all-ones V^T (every lane gets the complete sum).
// Lazy softmax reference with EARLY overflow votes: the first tile of a Q block takes the exact path and its row maxima become
// the reference m; every later unmasked tile is exponentiated against m AS IT STANDS - any reference within 2^15 of the row max
// is as good as the max (P is floating point, fp32 accumulate) - one fma + one exp per score and no row max.  A row whose scores
// outgrow m by more than 2^16 overflows fp16 P: the infinity reaches its row sum.  The waves look at their row sums after 2, 4,
// 8, ... tiles and at the end; when one reports an overflow the workgroup replays from tile 0 in exact mode (lane-partial
// integer maxima every tile, cross-lane step and rescale only where a row outgrew its reference by 2^8): the waves that
// overflowed start over, the others keep their accumulators and only pass the barriers until the replay reaches the point they
// had come to ("tile loop" below).  The same votes carry a second condition for the int8 kernels: a wave one of whose references
// has left +-2^7 binades - the range in which the rounded dequantisation scale of the one-fma form is exact enough (kGridRef) -
// reports too, and replays with the un-rounded scale (`wide`), as the reference dequantises (attn_qk_int8_per_block.py:51).
// LDS images (checked conflict-free by enumeration of the hardware's lane groups):
//   K tile [64][RB bytes]: 16-byte chunk c of row r at c ^ kx16(r), kx16 = (r >> 1) & 3 | r & 7 | r & 15 for RB = 64 | 128 | 256
//   V tile [64][2 D bytes]: 32-byte block c of row r at c ^ vx16(r), vx16 = (r >> 1) & 3 (D = 64) | r & 7 (D = 128)
// fp8 PV (one operand = all 64 keys of a tile) stays on the 32x32x64 block-scaled MFMA in attn_fwd.hip.
#include "attn_common.h"

namespace lbfa {

constexpr float kLazyThr = 8.0f;  // exact paths move the softmax reference only when a row max outgrows it by more than 2^8
// The one-fma dequantisation (scale rounded by <= 2^-19 relative onto the bias grid) moves the exponent of a score near the row max by
// up to |m| 2^-19: below 2^-12 - a fraction of the fp16 rounding of P - while the softmax reference m stays within 2^7 binades of
// zero.  A wave whose reference has left that range votes for the replay like one whose row sums overflowed, and replays with the
// UN-rounded scale and an exact bias subtraction (`wide`); a reference that leaves the range DURING a replay (only exact-path tiles
// move references) switches its wave to `wide` in that very tile.
constexpr float kGridRef = 128.0f;
constexpr bool kPingPong = true;  // every other round of Q blocks walks the key tiles backwards (L2 reuse, see attn_fwd.hip)
// V^T fragments are read in batches of kVBatch channel blocks (4 registers each), kVAhead batches ahead of the MFMAs that use
// them (kVAhead + 1 register sets).  Measured (S16K / D128 / C3): one block ahead at 1 / 2 blocks per batch +1 / +3..5 / +2 % over
// un-pipelined batches of 2 / 4; two or three batches ahead, or bigger batches (spills), no better.
template <int D> constexpr int kVBatch = (D == 64) ? 1 : 2;
constexpr int kVAhead = 1;

template <int RB>
__device__ __forceinline__ int kx16(int row) {  // K-tile 16-byte chunk swizzle, rows of RB bytes
  if constexpr (RB == 64) return (row >> 1) & 3;
  else if constexpr (RB == 128) return row & 7;
  else return row & 15;
}
template <int D>
__device__ __forceinline__ int vx16(int row) {  // V-tile 32-byte block swizzle
  if constexpr (D == 64) return (row >> 1) & 3;
  else return row & 7;
}
#if defined(LBFA_STAMPS16) && LBFA_D16 == LBFA_STAMPS16  // diagnostic build only (-DLBFA_STAMPS16=64 | 128, tools/stamps.py): s_memtime at points of a workgroup's life, wave 0 lane 0
// record per workgroup: [0..7] phases of the workgroup's life, [8..15] progress of wave 0's instruction stream through ONE tile of
// the lazy main loop (tile n_main / 2): step top, tile fetch issued, QK^T issued, exponentials of k-step 0 issued, PV k-step 0 +
// exponentials of k-step 1 issued, PV k-step 1 issued, prefetch landed (vmcnt 0), barrier passed
__device__ long long g_stamps16[8192 * 24];  // [16], [17]: s_memrealtime (100 MHz) at the start / end of the tile loop -> in-kernel clock
#define LBFA_STAMP(k)                                                                                              \
  do {                                                                                                             \
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_stamps16[blockIdx.x * 24 + (k)] = __builtin_amdgcn_s_memtime();   \
  } while (0)
#define LBFA_RSTAMP(k)                                                                                                 \
  do {                                                                                                                 \
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_stamps16[blockIdx.x * 24 + (k)] = __builtin_amdgcn_s_memrealtime();   \
  } while (0)
#define LBFA_TSTAMP(k)                                      \
  do {                                                      \
    __builtin_amdgcn_sched_barrier(0);                      \
    if (ts_on) ts[k] = __builtin_amdgcn_s_memtime();        \
    __builtin_amdgcn_sched_barrier(0);                      \
  } while (0)
extern "C" int lbfa_debug_stamps(void* dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps16), sizeof(g_stamps16), 0, hipMemcpyDeviceToHost);
}
#else
#define LBFA_STAMP(k)
#define LBFA_RSTAMP(k)
#define LBFA_TSTAMP(k)
#endif

template <int D, int QT, int VT, int OT, bool CAUSAL, bool QQ = false>
__global__ __launch_bounds__(256, D == 64 ? 3 : 2) void attn_fwd16_kernel(AttnParams p) {
  static_assert(!QQ || QT == kQInt8, "in-kernel Q quantisation belongs to the int8 path");
  static_assert(VT != LBFA_E4M3, "fp8 PV runs in attn_fwd.hip");
  constexpr bool QK16 = (QT != kQInt8);
  constexpr int ESZ = QK16 ? 2 : 1;        // bytes per Q / K element
  constexpr int RB = D * ESZ;              // bytes per K row
  constexpr float THR = kLazyThr;
  constexpr int KS = RB / 64;              // k-steps of the score product: 64 row bytes per MFMA (64 int8 or 32 fp16)
  constexpr int CB = D / 16;               // 16-channel blocks of O^T
  constexpr int KBYTES = 64 * RB, VBYTES = 128 * D;
  constexpr int KCH = KBYTES / 4096, VCH = VBYTES / 4096;  // 16-byte chunks per thread
  // P and V of the PV product: fp16 x fp16 (the reference's `p.to(float16)` x `v.to(float16)`, attn_qk_int8_per_block.py:59-61,
  // src/core.py:307-308: a bf16 V reaches the int8 operators already cast, lbfa_cast_bf16_to_f16) for the int8 operators; the
  // un-quantised bf16 kernel keeps both in bf16 (as a bf16 FlashAttention-2 does).  V tiles always arrive by LDS-DMA.
  constexpr bool PV_BF16 = (QT == LBFA_BF16);
  static_assert(VT == (PV_BF16 ? LBFA_BF16 : LBFA_F16), "int8 operators and the fp16 kernel take fp16 V, the bf16 kernel bf16 V");
  constexpr int TILES_BYTES = 2 * (KBYTES + VBYTES);
  __shared__ __attribute__((aligned(16))) char smem[TILES_BYTES + 16];  // ONE LDS object (see attn_fwd.hip)

  LBFA_STAMP(0);
  [[maybe_unused]] long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  [[maybe_unused]] bool ts_on = false;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int i16 = lane & 15, g = lane >> 4;

  // ---- which (batch, head, q-block) -----------------------------------------------------------------
  const unsigned w_id = xcd_remap(blockIdx.x, gridDim.x);
  int qt = (int)(w_id % (unsigned)p.nQ);
  const int bh = (int)(w_id / (unsigned)p.nQ);
  if constexpr (CAUSAL) qt = p.nQ - 1 - qt;  // heaviest q-blocks of a head first
  const int b = bh / p.Hq, h = bh % p.Hq, hk = h / p.group;

  int Sq = p.Sq, Sk = p.Sk, nK = p.nK;
  int64_t q_off = (int64_t)b * p.qb, k_off = (int64_t)b * p.kb, v_off = (int64_t)b * p.vb, o_off = (int64_t)b * p.ob;
  int64_t qsc_base = (int64_t)b * p.qsc_b, ksc_base = (int64_t)b * p.ksc_b;
  if (p.cu_q != nullptr) {  // packed variable-length batch (attn_qk_int8_block_varlen.py:125-141); lengths cut at the maxima
    const int q0 = p.cu_q[b], k0 = p.cu_k[b];
    Sq = min(p.cu_q[b + 1] - q0, p.Sq);
    Sk = min(p.cu_k[b + 1] - k0, p.Sk);
    if (qt * 128 >= Sq) return;  // whole workgroup, before any barrier
    nK = (Sk + 63) >> 6;
    q_off = (int64_t)q0 * p.qs;
    k_off = (int64_t)k0 * p.ks;
    v_off = (int64_t)k0 * p.vs;
    o_off = (int64_t)q0 * p.os;
    if (p.cu_qscale != nullptr) {
      qsc_base = (int64_t)p.cu_qscale[b] * p.qsc_b;
      ksc_base = (int64_t)p.cu_kscale[b] * p.ksc_b;
    }
  }
  const int row0 = qt * 128 + wave * 32;  // first query row of this wave
  auto qrow_of = [&](int rb) __attribute__((always_inline)) { return row0 + 16 * rb + i16; };

  // ---- Q rows of this wave: requested FIRST, in flight while the K / V fetch is set up and issued ----------------------
  // lane (i, g) takes row bytes [64 s + 16 g, +16) of the int8 codes / 16-bit elements (QQ: the 16 fp16 source elements
  // [64 s + 16 g, +16) as two 16-byte pieces) of query rows 16 rb + i; rows >= Sq are out of the descriptor's range: zeros
  constexpr int QPC = QQ ? 2 : 1;
  u32x4 qraw[2][KS][QPC];
  {
    constexpr int QESZ = QQ ? 2 : ESZ;  // bytes per element of the Q operand as stored
    const int q_valid = (QQ || QK16) ? p.d_valid : D;
    const char* qbase = (const char*)p.q + QESZ * (q_off + (int64_t)h * p.qh);
    const __amdgpu_buffer_rsrc_t q_rs = make_rsrc(qbase, (unsigned)(QESZ * ((int64_t)(Sq - 1) * p.qs + q_valid)));
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int hf = 0; hf < QPC; ++hf) {
          const unsigned col_b = QQ ? 2 * (64 * s + 16 * g + 8 * hf) : 64 * s + 16 * g;  // byte column
          qraw[rb][s][hf] = buf_load16(q_rs, col_b < (unsigned)(QESZ * q_valid) ? QESZ * (unsigned)qrow_of(rb) * (unsigned)p.qs + col_b : 0x80000000u, 0);
        }
  }

  // ---- operand windows (bytes) ------------------------------------------------------------------------
  const int dq_valid = QK16 ? p.d_valid : D;
  const char* kbase = (const char*)p.k + ESZ * (k_off + (int64_t)hk * p.kh);
  const int64_t k_bytes = ESZ * ((int64_t)(Sk - 1) * p.ks + dq_valid);
  const int64_t k_tile_stride = ESZ * 64 * p.ks;
  const char* vbase = (const char*)p.v + 2 * (v_off + (int64_t)hk * p.vh);
  const int64_t v_bytes = 2 * ((int64_t)(Sk - 1) * p.vs + p.d_valid);
  const int64_t v_tile_stride = 128 * p.vs;

  // ---- loop-invariant per-thread offsets of the tile fetch (one 16-byte chunk per thread and pass of 256 threads) ----
  constexpr int KCPR = RB / 16, KROWS = 256 / KCPR;
  constexpr int VCPR = D / 8, VROWS = 256 / VCPR;
  unsigned k_goff, v_goff;
  {
    const int row = t / KCPR, ch = t % KCPR;
    const int gch = ch ^ kx16<RB>(row);  // LDS-DMA writes linearly: the slot (row, ch) holds global chunk ch ^ kx16(row)
    k_goff = gch * 16 < ESZ * dq_valid ? ESZ * (unsigned)row * (unsigned)p.ks + gch * 16 : 0x80000000u;
  }
  {
    const int row = t / VCPR, ch = t % VCPR;
    const int gch = (((ch >> 1) ^ vx16<D>(row)) << 1) | (ch & 1);  // LDS-DMA writes linearly: the swizzle moves to the source
    v_goff = gch * 8 < p.d_valid ? 2 * ((unsigned)row * (unsigned)p.vs) + gch * 16 : 0x80000000u;
  }
  const unsigned k_gstep = ESZ * KROWS * (unsigned)p.ks;
  const unsigned v_gstep = 2u * VROWS * (unsigned)p.vs;
  static_assert(KROWS * RB == 4096 && VROWS * 2 * D == 4096, "one pass of 256 threads x 16 bytes");
  const int k_bytes32 = (int)k_bytes, v_bytes32 = (int)v_bytes, k_stride32 = (int)k_tile_stride, v_stride32 = (int)v_tile_stride;
  typedef __attribute__((address_space(3))) void* lds_void_ptr;
  auto load_tile = [&](int j, auto buf_tag) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_tag)::value;
    const bool in_range = (unsigned)j < (unsigned)nK;  // the look-ahead past either end gets a window of 0 bytes
    const int ko = in_range ? j * k_stride32 : 0, vo = in_range ? j * v_stride32 : 0;
    const int k_rem = in_range ? max(0, k_bytes32 - ko) : 0, v_rem = in_range ? max(0, v_bytes32 - vo) : 0;
    const __amdgpu_buffer_rsrc_t k_rs = make_rsrc(kbase + ko, (unsigned)k_rem);
    const __amdgpu_buffer_rsrc_t v_rs = make_rsrc(vbase + vo, (unsigned)v_rem);
    char* kdst = smem + BUF * KBYTES + wave * 1024;  // DMA destination: wave-uniform base (+ 16 bytes per lane, implicit)
#pragma unroll
    for (int c = 0; c < KCH; ++c)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rs, (lds_void_ptr)(kdst + c * 4096), 16, (int)k_goff, (int)(c * k_gstep), 0, 0);
    char* vdst = smem + 2 * KBYTES + BUF * VBYTES + wave * 1024;
#pragma unroll
    for (int c = 0; c < VCH; ++c)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rs, (lds_void_ptr)(vdst + c * 4096), 16, (int)v_goff, (int)(c * v_gstep), 0, 0);
  };

  // processing order of the key tiles (ping-pong per round of Q blocks, see attn_fwd.hip)
  constexpr int kRound = (D == 64) ? 96 : 64;
  const bool rev = !CAUSAL && kPingPong && ((Sk & 63) == 0) && (((qt / kRound) & 1) != 0);
  auto tile_of = [&](int i) __attribute__((always_inline)) { return rev ? nK - 1 - i : i; };
  load_tile(tile_of(0), std::integral_constant<int, 0>{});
  const float* ksc = nullptr;
  if constexpr (!QK16) ksc = p.k_scale + ksc_base + (int64_t)hk * p.ksc_h;
  const int ksc_blk = (int)p.ksc_blk;
  float ks_first = 0.f;
  if constexpr (!QK16) ks_first = lane < nK ? ksc[lane * ksc_blk] : 0.f;

  // ---- Q fragments (B operand of the score MFMAs) from the rows requested above --------------------------------------
  i32x4 qf[2][KS];
  float qsc = 1.0f;
  float row_corr[2] = {0.f, 0.f};
  if constexpr (QQ) {
    // in-kernel Q quantiser: same arithmetic as quant_per_block_kernel (src/triton/quant_per_block.py:132-178)
    float xs[2][KS][16];
    float amax = 0.f;
    const unsigned short* vec = p.q_dot_vec ? p.q_dot_vec + ((int64_t)b * p.Hkv + hk) * D : nullptr;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      float dot = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int col = 64 * s + 16 * g + 8 * hf;
          const u32x4 raw = qraw[rb][s][hf];
          float xv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            xv[e] = load_cvt<OT>((unsigned short)((e & 1) ? (raw[e >> 1] >> 16) : (raw[e >> 1] & 0xffffu)));
            const float x = xv[e] * p.q_sm_scale;
            xs[rb][s][8 * hf + e] = x;
            amax = fmaxf(amax, fabsf(x));
          }
          if (vec != nullptr) {
            const u32x4 vraw = *reinterpret_cast<const u32x4*>(vec + col);
#pragma unroll
            for (int e = 0; e < 8; ++e)
              dot += xv[e] * load_cvt<OT>((unsigned short)((e & 1) ? (vraw[e >> 1] >> 16) : (vraw[e >> 1] & 0xffffu)));
          }
        }
      row_corr[rb] = load_cvt<OT>(store_cvt<OT>(rows4_sum(dot)));  // rounded to the storage dtype (src/core.py:294-304)
    }
    amax = wave_max_nonneg(amax);
    LBFA_STAMP(6);
    float* red = reinterpret_cast<float*>(smem + TILES_BYTES);
    if (lane == 0) red[wave] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    LBFA_STAMP(7);
    const float scale = fmaxf(amax, 1e-7f) / p.q_qmax;
    qsc = scale;
    const float rcp = 1.0f / scale;
    const bool exact_rcp_ok = (__builtin_amdgcn_readfirstlane(__float_as_uint(scale)) & 0x7fffffu) != 0x7fffffu;
    auto encode = [&](auto fast_tag) __attribute__((always_inline)) {
      constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          unsigned w[4];
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            int qv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float xv = xs[rb][s][4 * g4 + e];
              float y;
              if constexpr (FAST) {
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
          qf[rb][s] = i32x4{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
        }
    };
    if (exact_rcp_ok) encode(std::true_type{});
    else encode(std::false_type{});
  } else {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int s = 0; s < KS; ++s) qf[rb][s] = __builtin_bit_cast(i32x4, qraw[rb][s][0]);
    if constexpr (!QK16) qsc = p.q_scale[qsc_base + (int64_t)h * p.qsc_h + (int64_t)qt * p.qsc_blk];
  }

  LBFA_STAMP(1);
  int n_tiles = nK;
  if constexpr (CAUSAL) n_tiles = min(nK, 2 * (qt + 1));

  // ---- fragment read addresses (lane parts; block / k-step / buffer parts are immediates) ------------------------
  const unsigned kf_lane = i16 * RB + ((g ^ kx16<RB>(i16)) << 4);  // k-step s: ^ (s << 6)
  unsigned vf_base[CB];
  {
    const int vr = 4 * g + (i16 >> 2);  // row within a 16-key group of the tile (+ 32 s + 16 half as immediates)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) vf_base[cb] = 2 * KBYTES + vr * (2 * D) + ((cb ^ vx16<D>(vr)) << 5) + (i16 & 3) * 8;
  }
  // V fragments are read by hand-issued ds_read_b64_tr_b16 (attn_common.h, lds_read_tr16_raw): the compiler then orders nothing
  // against the LDS-DMA prefetch in flight, so the tile loop waits for it itself - vmcnt(0) in front of each barrier
  unsigned vf_addr[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) vf_addr[cb] = lds_offset_of(smem) + vf_base[cb];

  // ---- running state --------------------------------------------------------------------------------
  // Row sums.  The SIMD's vector issue port is the scarce resource of the tile loop (MI355X_MICROARCH.md, per-instruction issue
  // costs: plain VALU 4 cycles, v_exp_f32 8, an MFMA 8), so the sums ride on the matrix pipe: O^T gets a (D/16 + 1)-th
  // channel block whose V^T rows are all ones - one more 16x16x32 MFMA per row block and k-step (8 issue cycles for 128
  // additions; 32 v_add_f32 would take 128) - and every accumulator element of that block IS the complete row sum of the lane's
  // query: no cross-lane step at the end either.
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  f16x8 ones8;  // the bit patterns of eight ones in the PV dtype
#pragma unroll
  for (int e = 0; e < 8; ++e) ones8[e] = __builtin_bit_cast(_Float16, (unsigned short)(PV_BF16 ? 0x3F80 : 0x3C00));
  asm volatile("" : "+v"(ones8));  // opaque: otherwise re-materialised in every tile
  // one PV MFMA: A = V^T (or the ones), B = P^T, both as raw 16-bit fragments
  auto pv_mfma = [&](f16x8 a, f16x8 b, f32x4 c) __attribute__((always_inline)) {
    if constexpr (PV_BF16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  };
  f32x4 acc_o[2][CB];
  f32x4 l_acc[2];
  float m_run[2];
  auto reset_state = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) acc_o[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
      l_acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
      m_run[rb] = -INFINITY;
    }
  };
  reset_state();
  i32x4 cmagic = i32x4{kMagicBits, kMagicBits, kMagicBits, kMagicBits};
  asm volatile("" : "+v"(cmagic));

  // ---- exact bias folding (see attn_fwd.hip): constants on a common power-of-two grid ------------------------------
  float ks_max = 0.f;
  if constexpr (!QK16) {
    ks_max = ks_first;
    for (int i = lane + 64; i < nK; i += 64) ks_max = fmaxf(ks_max, ksc[i * ksc_blk]);
    ks_max = fmaxf(wave_max_nonneg(ks_max), 1e-30f);
  }
  // wave-uniform values below are moved to SGPRs by hand (the compiler keeps uniform floats in VGPRs, and the tile loop has none
  // to spare: left alone, the QQ instances parked 14 cold dwords in scratch)
  auto uniform = [](float v) __attribute__((always_inline)) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
  };
  qsc = uniform(qsc);
  const float sc_max = qsc * ks_max;
  const int gexp = __builtin_amdgcn_readfirstlane((int)((__float_as_uint(1.25f * kMagic * sc_max) >> 23) & 0xff) - 127 + 1 - 21);  // log2(G)
  const float G = uniform(__builtin_ldexpf(1.0f, gexp)), invG = uniform(__builtin_ldexpf(1.0f, -gexp));
  const float gg = uniform(__builtin_ldexpf(1.0f, gexp - 22)), invg = uniform(__builtin_ldexpf(1.0f, 22 - gexp));
  // Wide scores: the grid is one integer step of a score wide at best (G in (7.5, 15] sc_max), and a reference rounded up to it puts
  // the largest P at 2^-G.  Harmless while a step is a fraction of a binade; inputs whose 8-bit (4-bit) steps are binades apart -
  // the reference's bench distribution randint(-100, 100) on the 4-bit-range codes: G = 256 - take the bias off the scores with one
  // exact subtraction each instead (tv - kMagic = s), keep the scales and the reference unrounded, and pay 32 VALU per tile.
  // The same arithmetic takes over, per wave, when one of its softmax references leaves +-kGridRef (vote -> replay, or inside a replay:
  // compute_tile, exact path): the reference dequantises in fp32 (attn_qk_int8_per_block.py:51), and on scores thousands of binades
  // wide - its own bench distribution randint(-100, 100) on 8-bit codes, G = 1..2, references ~5000 - a scale rounded by 2^-19 shifts
  // the weights of keys whose scores tie across tiles.
  const bool wide_all = !QK16 && gexp >= 2;
  bool wide = wide_all;  // wave-uniform; only ever switched on
  auto grid_up = [&](float m) __attribute__((always_inline)) { return wide ? m : __builtin_ceilf(m * invG) * G; };
  // per lane: tile 64 c + lane's dequantisation scale as the tile loop uses it (on the grid, or un-rounded for a `wide` wave) and its
  // raw k_scale (what a wave that goes `wide` in the middle of a replay rebuilds the first from, without a memory access)
  float sc_tab = 0.f, ks_tab = 0.f;
  auto refresh_scale_table = [&](int j0) __attribute__((always_inline)) {
    if constexpr (!QK16) {
      const int jt = j0 + lane;
      const float ks_l = j0 == 0 ? ks_first : (jt < nK ? ksc[jt * ksc_blk] : 0.f);
      ks_tab = ks_l;
      sc_tab = fmaxf(__builtin_rintf(qsc * ks_l * invg), 1.0f) * gg;  // >= one grid step: a masked key must not meet sc = 0
      if (wide) sc_tab = fmaxf(qsc * ks_l, 1e-30f);
    }
  };

  auto off_grid = [&]() __attribute__((always_inline)) {  // has a reference of this wave left the range?  (-inf: a row with no key yet)
    const float r0 = __builtin_fabsf(m_run[0]), r1 = __builtin_fabsf(m_run[1]);
    return __any((r0 > kGridRef && r0 < INFINITY) || (r1 > kGridRef && r1 < INFINITY)) != 0;
  };
  // replay_tag: this tile belongs to a replay (the lazy pass leaves the switch to `wide` to its votes: its tile bodies carry no code for it)
  auto compute_tile = [&](auto buf_tag, int j, auto masked_tag, auto exact_tag, auto replay_tag) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_tag)::value;
    constexpr bool MASKED = decltype(masked_tag)::value;
    constexpr bool EXACT = decltype(exact_tag)::value || MASKED;
    constexpr bool REPLAY = decltype(replay_tag)::value;
    const char* kbuf = smem + BUF * KBYTES;
    const float bias = wide ? kMagic : 0.f;
    float sc, c0;
    if constexpr (QK16) {
      sc = p.qk_scale;
      c0 = 0.f;
    } else {
      sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc_tab), j & 63));
      c0 = wide ? 0.f : -kMagic * sc;  // exact on the grid (sc = k G / 2^22: 3 k G)
    }
    float x[2][4][4];  // [row block][key block][key 4 g + e]: kMagic + s (accumulator bits), then P in place
    auto compute_scores = [&](auto kb_tag) __attribute__((always_inline)) {
      constexpr int kb = decltype(kb_tag)::value;
      i32x4 sacc[2];
      f32x4 facc[2];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const i32x4 kf = *reinterpret_cast<const i32x4*>(kbuf + (kf_lane ^ (unsigned)(s << 6)) + kb * 16 * RB);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          if constexpr (QT == LBFA_BF16) {
            const bf16x8 ka = __builtin_bit_cast(bf16x8, kf), qb = __builtin_bit_cast(bf16x8, qf[rb][s]);
            if (s == 0) facc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qb, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            else facc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qb, facc[rb], 0, 0, 0);
          } else if constexpr (QK16) {
            const f16x8 ka = __builtin_bit_cast(f16x8, kf), qb = __builtin_bit_cast(f16x8, qf[rb][s]);
            if (s == 0) facc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, qb, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            else facc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, qb, facc[rb], 0, 0, 0);
          } else {
            if (s == 0) sacc[rb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(kf, qf[rb][s], cmagic, 0, 0, 0);
            else sacc[rb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(kf, qf[rb][s], sacc[rb], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float tv;
          if constexpr (QK16) tv = facc[rb][e];
          else tv = __int_as_float(sacc[rb][e]);
          if constexpr (MASKED) {
            const int key = j * 64 + 16 * kb + 4 * g + e;
            bool dead = key >= Sk;
            if constexpr (CAUSAL) dead = dead || (key > qrow_of(rb));
            if (dead) tv = -INFINITY;
          }
          x[rb][kb][e] = tv;
        }
    };
    // Lane-partial maximum of the 16 scores of a row block, as an order key (attn_common.h): 7 v_max3 + 1 v_max.
    auto lane_max = [&](int rb) __attribute__((always_inline)) {
      float m = key_max3<!QK16>(x[rb][0][0], x[rb][0][1], x[rb][0][2]);
      m = key_max3<!QK16>(m, x[rb][0][3], x[rb][1][0]);
      m = key_max3<!QK16>(m, x[rb][1][1], x[rb][1][2]);
      m = key_max3<!QK16>(m, x[rb][1][3], x[rb][2][0]);
      m = key_max3<!QK16>(m, x[rb][2][1], x[rb][2][2]);
      m = key_max3<!QK16>(m, x[rb][2][3], x[rb][3][0]);
      m = key_max3<!QK16>(m, x[rb][3][1], x[rb][3][2]);
      return key_max<!QK16>(m, x[rb][3][3]);
    };
    // Move the softmax reference of row block rb up to this tile's row max where a row outgrew it by more than thr, rescaling
    // what has been accumulated against the old one (the rare path: cross-lane maximum, grid rounding, O-wide multiply).
    auto raise_reference = [&](int rb, float pm, float thr) __attribute__((always_inline)) {
      const float tmax = rows4_key_max<!QK16>(pm);
      const float xmax = __builtin_fmaf(QK16 ? tmax : tmax - bias, sc, c0);
      const float m_cand = fmaxf(m_run[rb], QK16 ? xmax : grid_up(xmax));
      if (__any(m_cand > m_run[rb] + thr)) {
        const float alpha = __builtin_amdgcn_exp2f(m_run[rb] - m_cand);  // m_run = -inf -> 0
        m_run[rb] = m_cand;
#pragma unroll
        for (int e = 0; e < 4; ++e) l_acc[rb][e] *= alpha;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc_o[rb][cb][e] *= alpha;
      }
    };
    f16x8 pf[2][2];  // [row block][k-step of 32 keys]
    float c1[2];
    // P of one k-step (key blocks 2 s, 2 s + 1) of one row block, in place, + the packed P^T fragment
    auto exp_s = [&](auto rb_tag, auto s_tag) __attribute__((always_inline)) {
      constexpr int rb = decltype(rb_tag)::value, s = decltype(s_tag)::value;
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          x[rb][2 * s + k2][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[rb][2 * s + k2][e], sc, c1[rb]));
        }
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (PV_BF16) pf[rb][s][4 * k2 + e] = __builtin_bit_cast(_Float16, (__bf16)x[rb][2 * s + k2][e]);
          else pf[rb][s][4 * k2 + e] = (_Float16)x[rb][2 * s + k2][e];
        }
    };
    // O^T += V^T P^T.  The V^T fragments of a tile are read in NBT = 2 CB / VG batches of VG channel blocks (every fragment
    // serves both row blocks); batch b + VD is requested before the MFMAs of batch b are issued (VD + 1 register sets, counted
    // lgkmcnt waits: LDS returns in order and the wave has no other LDS traffic in this phase), the first VD batches of the tile
    // before the exponentials.
    constexpr int VG = kVBatch<D>;
    constexpr int NQ = CB / VG, NBT = 2 * NQ;
    constexpr int VD = kVAhead;  // batches requested ahead of the one whose MFMAs are being issued
    f16x4 vb_lo[VD + 1][VG], vb_hi[VD + 1][VG];
    auto v_issue = [&](auto b_tag) __attribute__((always_inline)) {
      constexpr int bb = decltype(b_tag)::value, s = bb / NQ, c0 = VG * (bb % NQ), slot = bb % (VD + 1);
      constexpr int off = BUF * VBYTES + (32 * s) * (2 * D);
      static_for<0, VG>([&](auto c) {
        constexpr int ci = decltype(c)::value;
        vb_lo[slot][ci] = lds_read_tr16_raw<off>(vf_addr[c0 + ci]);
        vb_hi[slot][ci] = lds_read_tr16_raw<off + 16 * 2 * D>(vf_addr[c0 + ci]);
      });
    };
    auto v_use = [&](auto b_tag) __attribute__((always_inline)) {
      constexpr int bb = decltype(b_tag)::value, s = bb / NQ, c0 = VG * (bb % NQ), slot = bb % (VD + 1);
      constexpr int AHEAD = (NBT - 1 - bb) < VD ? (NBT - 1 - bb) : VD;
      constexpr int KEEP = 2 * VG * AHEAD;  // reads of the following batches may stay in flight
      if constexpr (VG == 4) lds_wait_keep<KEEP>(vb_lo[slot][0], vb_hi[slot][0], vb_lo[slot][1], vb_hi[slot][1], vb_lo[slot][2], vb_hi[slot][2], vb_lo[slot][3], vb_hi[slot][3]);
      else if constexpr (VG == 2) lds_wait_keep<KEEP>(vb_lo[slot][0], vb_hi[slot][0], vb_lo[slot][1], vb_hi[slot][1]);
      else lds_wait_keep<KEEP>(vb_lo[slot][0], vb_hi[slot][0]);
      static_for<0, VG>([&](auto c) {
        constexpr int ci = decltype(c)::value, cb = c0 + ci;
        const f16x4 lo = vb_lo[slot][ci], hi = vb_hi[slot][ci];
        const f16x8 vf = f16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) acc_o[rb][cb] = pv_mfma(vf, pf[rb][s], acc_o[rb][cb]);
      });
    };
    auto pv_s = [&](auto s_tag) __attribute__((always_inline)) {
      constexpr int s = decltype(s_tag)::value;
      static_for<0, NQ>([&](auto q) {
        constexpr int bb = s * NQ + decltype(q)::value;
        if constexpr (bb + VD < NBT) v_issue(std::integral_constant<int, bb + VD>{});
        v_use(std::integral_constant<int, bb>{});
      });
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) l_acc[rb] = pv_mfma(ones8, pf[rb][s], l_acc[rb]);
    };
    using R0 = std::integral_constant<int, 0>;
    using R1 = std::integral_constant<int, 1>;

    static_for<0, 4>([&](auto kb) { compute_scores(kb); });
    LBFA_TSTAMP(2);
    c1[0] = c0 - m_run[0];  // exact (grid argument); +inf while m_run = -inf
    c1[1] = c0 - m_run[1];
    if constexpr (EXACT) {
      // Exact path: every tile tests its LANE-partial maxima against the reference - the exponent argument of the largest score
      // the lane holds, 2 x (8 v_max + 1 v_fma + 1 v_cmp) - and only a wave that finds a row more than 2^THR above its reference
      // takes the cross-lane step and the rescale.
      const float pm0 = lane_max(0), pm1 = lane_max(1);
      const float a0 = __builtin_fmaf(QK16 ? pm0 : pm0 - bias, sc, c1[0]), a1 = __builtin_fmaf(QK16 ? pm1 : pm1 - bias, sc, c1[1]);
      if (__any((a0 > THR) || (a1 > THR))) {  // also the first tile (c1 = +inf)
        raise_reference(0, pm0, THR);
        raise_reference(1, pm1, THR);
        if constexpr (!QK16 && REPLAY) {
          if (!wide) {
            if (off_grid()) {
              // this tile and every later one of the wave: un-rounded scale, bias subtracted exactly (below).  Rare; only the tile
              // bodies of the replay carry this code, and it touches no memory: the table is rebuilt from the raw k_scales it keeps
              wide = true;
              sc_tab = fmaxf(qsc * ks_tab, 1e-30f);
              sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc_tab), j & 63));
              c0 = 0.f;
            }
          }
        }
        c1[0] = c0 - m_run[0];
        c1[1] = c0 - m_run[1];
      }
    }
    if constexpr (!QK16) {
      if (wide) {  // wave-uniform: scores as the integers themselves
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) x[rb][kb][e] -= kMagic;
      }
    }
    static_for<0, (VD < NBT ? VD : NBT)>([&](auto b0) { v_issue(b0); });
    exp_s(R0{}, R0{});
    exp_s(R1{}, R0{});
    LBFA_TSTAMP(3);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    pv_s(R0{});
    exp_s(R0{}, R1{});
    exp_s(R1{}, R1{});
    {
      constexpr int NM = 2 * CB + 2;  // MFMAs of k-step 0: PV + row sums
      constexpr int NV = 40 / NM;     // 16 fma + 16 exp + 8 cvt of k-step 1 spread over them
      static_for<0, NM>([&](auto) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
      });
    }
    __builtin_amdgcn_sched_barrier(0);
    LBFA_TSTAMP(4);
    pv_s(R1{});
    LBFA_TSTAMP(5);
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- tile loop -----------------------------------------------------------------------------------------------------
  // One barrier per tile, buffers alternate statically (loop unrolled by two).  Full tiles first, branch-free; then the at
  // most three tiles that need masking (causal diagonal block = 2 tiles, ragged last tile), which always run the exact path.
  //
  // Lazy mode (the first attempt at every Q block): the reference is the exact row max of the first tile, every later
  // unmasked tile is exponentiated against it as it stands.  A row whose scores outgrow it by more than 2^16 overflows fp16 P:
  // the infinity reaches its row sum.  The waves look at their row sums after 2, 4, 8, 16, ... tiles and at the end (a vote
  // through 16 bytes of LDS on the tile's own barrier).  When a wave reports an overflow, the workgroup goes back to tile 0 in
  // exact mode: the waves that overflowed start over; the others keep what they have, only pass the barriers and fetch
  // their share of the tiles until the replay reaches the point they had come to, and continue from there in exact mode.  The
  // waste is at most twice the position of the first overflow, for the waves that overflowed only.
  int n_main = n_tiles;
  if constexpr (CAUSAL) n_main = min(n_tiles, 2 * qt);
  else if ((Sk & 63) != 0) n_main = n_tiles - 1;
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  using No = std::false_type;
  using Yes = std::true_type;
  int* vote_flag = reinterpret_cast<int*>(smem + TILES_BYTES);
  // A wave votes for the replay when a row sum has overflowed (every element of the all-ones block is a complete row sum) - the
  // un-quantised bf16 kernel, whose P cannot overflow before the fp32 accumulators of O could, when one has passed 2^64 - or,
  // on the rounded-scale grid, when one of its softmax references has left [-kGridRef, kGridRef].
  auto wave_overflowed = [&]() __attribute__((always_inline)) {
    const float chk = l_acc[0][0] + l_acc[1][0];
    bool bad = __any(!(chk < (PV_BF16 ? 0x1p64f : INFINITY))) != 0;
    if constexpr (!QK16) {
      if (!wide) bad = bad || off_grid();
    }
    return bad ? 1 : 0;
  };
  int skip_until = 0;  // replay: tiles below this index are already in this wave's accumulators
  // returns (when `vote`) whether any wave of the workgroup has an overflowed row sum
  auto step = [&](auto buf_tag, auto nbuf_tag, int i, auto masked_tag, auto exact_tag, bool vote, auto replay_tag) __attribute__((always_inline)) {
    const int j = tile_of(i);
#if defined(LBFA_STAMPS16) && LBFA_D16 == LBFA_STAMPS16
    ts_on = !decltype(exact_tag)::value && i == (n_main >> 1) && wave == 0;
#endif
    LBFA_TSTAMP(0);
    if (i != 0 && (j & 63) == (rev ? 63 : 0)) refresh_scale_table(j & ~63);
    load_tile(tile_of(i + 1), nbuf_tag);
    LBFA_TSTAMP(1);
    bool skip = false;
    if constexpr (decltype(masked_tag)::value && CAUSAL) skip = j * 64 > row0 + 31;  // all keys above all rows of this wave
    if constexpr (decltype(exact_tag)::value) skip = skip || i < skip_until;
    if (!skip) compute_tile(buf_tag, j, masked_tag, exact_tag, replay_tag);
    if (vote) {
      const int bad = wave_overflowed();
      if (lane == 0) vote_flag[wave] = bad;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next tile has landed (this wave's share) when the barrier opens
    LBFA_TSTAMP(6);
    __syncthreads();
    LBFA_TSTAMP(7);
    int any_bad = 0;
    if (vote) {  // the next write to vote_flag is at least one barrier away
      const i32x4 f = *reinterpret_cast<const i32x4*>(vote_flag);
      any_bad = __builtin_amdgcn_readfirstlane(f[0] | f[1] | f[2] | f[3]);
    }
    return any_bad;
  };
  // tiles [i0, n_tiles), i0 even (tile i0 in buffer 0); lazy mode returns early with the tile count reached when a vote fails
  auto run_tiles = [&](auto exact_tag, int i0) __attribute__((always_inline)) {
    constexpr bool EX = decltype(exact_tag)::value;
    using RP = std::integral_constant<bool, EX>;  // exact mode from the first tile on = a replay
    int i = i0;
    if constexpr (!EX) {
      // lazy mode: the first tile takes the exact path - its row maxima become the reference - and every later full tile is
      // exponentiated against the reference as it stands
      if (n_main >= 2) {
        step(B0{}, B1{}, 0, No{}, Yes{}, false, RP{});
        if (step(B1{}, B0{}, 1, No{}, No{}, 2 < n_main, RP{})) return 2;
        i = 2;
      }
    }
    for (; i + 1 < n_main; i += 2) {
      step(B0{}, B1{}, i, No{}, exact_tag, false, RP{});
      const int d = i + 2;
      const bool vote = !EX && (d & (d - 1)) == 0 && d < n_main;
      if (step(B1{}, B0{}, i + 1, No{}, exact_tag, vote, RP{})) return d;
    }
    // i is even and at most one full tile is left (odd n_main): it takes the exact path in either mode (with n_main = 1 it is
    // the tile that sets the reference); every tile after it is masked
    for (; i < n_tiles; i += 2) {
      if (i < n_main) step(B0{}, B1{}, i, No{}, Yes{}, false, RP{});
      else step(B0{}, B1{}, i, Yes{}, Yes{}, false, RP{});
      if (i + 1 < n_tiles) step(B1{}, B0{}, i + 1, Yes{}, Yes{}, false, RP{});
    }
    return -1;
  };
  auto first_tile_landed = [&]() __attribute__((always_inline)) {
    refresh_scale_table(tile_of(0) & ~63);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  LBFA_STAMP(2);
  LBFA_RSTAMP(16);
  first_tile_landed();
  int replay_end = run_tiles(No{}, 0);
  LBFA_STAMP(3);
  LBFA_RSTAMP(17);
  int my_bad = 0;
  if (replay_end < 0) {  // the lazy pass came to the end: the last vote, on a barrier of its own
    my_bad = wave_overflowed();
    if (lane == 0) vote_flag[wave] = my_bad;
    __syncthreads();
    const i32x4 f = *reinterpret_cast<const i32x4*>(vote_flag);
    if (__builtin_amdgcn_readfirstlane(f[0] | f[1] | f[2] | f[3])) replay_end = n_tiles;
  } else {
    my_bad = wave_overflowed();
  }
  if (replay_end >= 0) {
    __syncthreads();  // every wave has read the flags and left the tile buffers
    if constexpr (!QK16) {
      if (!wide && off_grid()) wide = true;  // such a wave voted (my_bad): it starts over, un-rounded (first_tile_landed() rebuilds the table)
    }
    if (my_bad) reset_state();
    skip_until = my_bad ? 0 : replay_end;
    load_tile(tile_of(0), B0{});
    first_tile_landed();
    run_tiles(Yes{}, 0);
  }
  float l_tot[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) l_tot[rb] = l_acc[rb][0];

  LBFA_STAMP(4);
  // ---- epilogue: O = O^T / l, LSE ------------------------------------------------------------------------------
  // (8-byte stores per lane, the four g-lanes of a row on one 32-byte sector.  Staging O through LDS into whole-row dwordx4
  // stores measured no faster: C2 +0.1 %, causal S4K -0.7 %, D128 0 %.)
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int qrow = qrow_of(rb);
    const float inv_l = l_tot[rb] > 0.f ? 1.0f / l_tot[rb] : 0.f;
    if (qrow < Sq) {
      unsigned short* op = reinterpret_cast<unsigned short*>(p.o) + o_off + (int64_t)h * p.oh + (int64_t)qrow * p.os;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const int d0 = 16 * cb + 4 * g;
        if (d0 >= p.d_valid) continue;  // d_valid is a multiple of 8
        uint2 pk;
        pk.x = (unsigned)store_cvt<OT>(acc_o[rb][cb][0] * inv_l) | ((unsigned)store_cvt<OT>(acc_o[rb][cb][1] * inv_l) << 16);
        pk.y = (unsigned)store_cvt<OT>(acc_o[rb][cb][2] * inv_l) | ((unsigned)store_cvt<OT>(acc_o[rb][cb][3] * inv_l) << 16);
        *reinterpret_cast<uint2*>(op + d0) = pk;
      }
      if (p.lse != nullptr && g == 0) {
        float ls = log2f(l_tot[rb]) + m_run[rb];  // base-2 domain (attn_qk_int8_per_block.py:164-167)
        const int64_t li = ((int64_t)b * p.Hq + h) * p.Sq + qrow;
        ls *= p.lse_scale;
        if constexpr (QQ) ls += row_corr[rb] * p.lse_corr_scale;
        else if (p.lse_corr != nullptr) ls += p.lse_corr[li] * p.lse_corr_scale;
        p.lse[li] = ls;
      }
    }
  }
  LBFA_STAMP(5);
#if defined(LBFA_STAMPS16) && LBFA_D16 == LBFA_STAMPS16
  if (threadIdx.x == 0 && blockIdx.x < 8192)
    for (int k = 0; k < 8; ++k) g_stamps16[blockIdx.x * 24 + 8 + k] = ts[k];
#endif
}

// ---- launchers (called from attn_fwd.hip's launch_* for every fp16-P variant) ----------------------------------------
// This file is compiled TWICE (Makefile: -DLBFA_D16=64 -> attn_fwd16_d64.o, -DLBFA_D16=128 -> attn_fwd16_d128.o): each
// translation unit instantiates the kernels of one head dim and exports launch16_*_d64 / _d128 - the two halves build in parallel.
#ifndef LBFA_D16
#error "compile with -DLBFA_D16=64 or -DLBFA_D16=128"
#endif
#define LBFA_CAT2(a, b) a##b
#define LBFA_CAT(a, b) LBFA_CAT2(a, b)
#define LBFA_DNAME(fn) LBFA_CAT(fn, LBFA_D16)

// int8 Q and K codes, fp16 V (lbfa_attn_fwd: bf16 V is cast by the caller, as src/core.py:307-308 does before its kernel call)
hipError_t LBFA_DNAME(launch16_attn_fwd_d)(const AttnParams& p, int v_dtype, int o_dtype, int causal, hipStream_t stream) {
  if (v_dtype != LBFA_F16) return hipErrorInvalidValue;
  dim3 grid((unsigned)p.B * p.Hq * p.nQ), block(256);
#define LBFA_A(OT)                                                                                                    \
  do {                                                                                                                \
    if (causal) hipLaunchKernelGGL((attn_fwd16_kernel<LBFA_D16, kQInt8, LBFA_F16, OT, true>), grid, block, 0, stream, p);  \
    else hipLaunchKernelGGL((attn_fwd16_kernel<LBFA_D16, kQInt8, LBFA_F16, OT, false>), grid, block, 0, stream, p);        \
  } while (0)
  if (o_dtype == LBFA_F16) LBFA_A(LBFA_F16);
  else LBFA_A(LBFA_BF16);
#undef LBFA_A
  return hipGetLastError();
}

// Q quantised in the kernel from its fp16 / bf16 source (dtype, also O's); V fp16 (the one-call operators cast a bf16 V in a
// pre-pass: converting it on the way into LDS cost the kernel 9..10 % and 25..56 spilled registers)
hipError_t LBFA_DNAME(launch16_attn_fwd_qq_d)(const AttnParams& p, int dtype, int v_dtype, int causal, hipStream_t stream) {
  if (v_dtype != LBFA_F16) return hipErrorInvalidValue;
  dim3 grid((unsigned)p.B * p.Hq * p.nQ), block(256);
#define LBFA_QQ(DT)                                                                                                          \
  do {                                                                                                                       \
    if (causal) hipLaunchKernelGGL((attn_fwd16_kernel<LBFA_D16, kQInt8, LBFA_F16, DT, true, true>), grid, block, 0, stream, p); \
    else hipLaunchKernelGGL((attn_fwd16_kernel<LBFA_D16, kQInt8, LBFA_F16, DT, false, true>), grid, block, 0, stream, p);       \
  } while (0)
  if (dtype == LBFA_F16) LBFA_QQ(LBFA_F16);
  else LBFA_QQ(LBFA_BF16);
#undef LBFA_QQ
  return hipGetLastError();
}

hipError_t LBFA_DNAME(launch16_attn_fwd_f16_d)(const AttnParams& p, int dtype, int causal, hipStream_t stream) {
  dim3 grid((unsigned)p.B * p.Hq * p.nQ), block(256);
#define LBFA_F(DT)                                                                                             \
  do {                                                                                                         \
    if (causal) hipLaunchKernelGGL((attn_fwd16_kernel<LBFA_D16, DT, DT, DT, true>), grid, block, 0, stream, p);   \
    else hipLaunchKernelGGL((attn_fwd16_kernel<LBFA_D16, DT, DT, DT, false>), grid, block, 0, stream, p);         \
  } while (0)
  if (dtype == LBFA_F16) LBFA_F(LBFA_F16);
  else LBFA_F(LBFA_BF16);
#undef LBFA_F
  return hipGetLastError();
}

}  // namespace lbfa
