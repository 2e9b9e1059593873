// Fused low-bit FlashAttention-2 forward for gfx950 (CDNA4): INT8 MFMA for QK^T, FP16 (or FP8) MFMA for PV.
//
// Replaces `_attn_fwd` / `_attn_fwd_inner` of the reference's Triton path
// (src/triton/attn_qk_int8_per_block.py:24-167, ..._causal.py:24-214) and, for fp8 V, the arithmetic of
// csrc/qattn/qk_int_sv_f8_cuda.cu:46-692.  Nothing here is derived from those sources' structure; the
// tiling below is chosen for 64-wide wavefronts and the MFMA register layouts:
//
//  * one workgroup = 4 waves = one 128-row Q block (= one q_scale), each wave owns 32 query rows;
//  * K/V stream through LDS in 64-key tiles (= one k_scale each), double-buffered, staged through
//    registers with 16-byte buffer loads issued one tile ahead (they fly during the MFMAs).  Buffer
//    descriptors bound every operand to its valid extent, so ragged tails read as zeros with no
//    per-lane guards, and the per-tile address update is one scalar add (voffset is loop-invariant);
//  * the score product is computed TRANSPOSED, S^T = K Q^T with v_mfma_i32_32x32x32_i8, so that a
//    lane owns ONE query row (column of S^T = lane&31) and 16 keys per 32-key block in registers:
//    row max / row sum are in-lane reductions plus a single v_permlane32_swap across the two halves;
//  * the int32 scores never pass through v_cvt_f32_i32 (a half-rate VALU op on gfx950): the MFMA
//    accumulates on top of 1.5*2^23, whose bit pattern + s IS the float 12582912+s; subtracting the
//    row maximum (same bias) gives s - smax exactly, and the dequant scale q_scale*k_scale is folded
//    into the exp2 argument with one v_fma;
//  * S^T accumulators feed the PV product directly as the B operand of v_mfma_f32_32x32x16_f16
//    (O^T = V^T P^T): no LDS round trip for P.  V^T fragments come from a row-major V tile in LDS via
//    ds_read_b64_tr_b16 (hardware transpose);  O^T keeps the query row on the lane, so the online
//    softmax rescale is a per-lane scalar multiply, and it is deferred while no row's max grows by
//    more than 2^THR (P stays exactly representable).
//
// LDS images (bank-conflict-free for the access patterns above):
//   K tile  [64 keys][D bytes]   16-B chunk c of row r stored at chunk c ^ kx(r)
//   V tile  [64 keys][D fp16]    64-B chunk c of row r stored at chunk c ^ vx(r)
//   V fp8   [D][64 bytes]        keys permuted into MFMA k order and 16-B chunks swizzled in HBM by lbfa_quant_v_fp8 -> linear copy
#include "attn_common.h"

namespace lbfa {

#ifndef LBFA_PRIO
#define LBFA_PRIO 2  // s_setprio(1) around the PV MFMA section: keeps the matrix pipe fed while other waves exponentiate (+3..6 %)
#endif
#ifndef LBFA_PRIO_FP8
#define LBFA_PRIO_FP8 0  // fp8 PV (2 / 4 long block-scaled MFMAs per tile): measured best without it (+2.7 % C5, +4 % D=64)
#endif
// fp16 P row sums: 0 = fp32 v_add on the VALU, 2 = on the matrix pipe (v_mfma_f32_4x4x4_16b_f16 with an all-ones A operand, each lane
// sums its own values).  Measured (C2 / S=16K / D=128 / C3, TFLOP/s, same box): VALU 980 / 1120 / 1214 / 1289, MFMA 990 / 1137 /
// 1190 / 1286, all-ones 32x32x16 MFMA (16 accumulator registers, two waves per SIMD at D = 64) 853 / 996: MFMA sums at D = 64
// where the VALU is the busier pipe, VALU sums at D = 128 where the matrix pipe is (re-measured on the interleaved D = 128
// path: +1 % at S = 4K, 0 at C3 - left on the VALU).
#ifndef LBFA_LSUM128
#define LBFA_LSUM128 0
#endif
#define LBFA_LSUM(D) ((D) == 64 ? 2 : LBFA_LSUM128)
// Interleave inside the tile (sched_group_barrier): 0 = one long VALU phase, then the PV MFMAs under s_setprio; 1 = the PV
// MFMAs of the first 32-key block pinned between the exponentials of the second; 2 = per 16-key k-step (PV of step r - 1
// between the exponentials of step r).  Measured against 0 / 1: D = 64 +4 % / +1 % for 2; D = 128: -1..-3 % without, 0..+2 %
// with the raised priority over the interleaved region (LBFA_ILV_PRIO).
#ifndef LBFA_ILV64
#define LBFA_ILV64 2
#endif
#ifndef LBFA_ILV128
#define LBFA_ILV128 2
#endif
#define LBFA_ILV(D) ((D) == 64 ? LBFA_ILV64 : LBFA_ILV128)
// Non-causal: every other ROUND of Q blocks of a head (a round = the workgroups one XCD runs at a time) walks the key tiles
// from the last one down.  The K + V panel of a head (6..8 MB at S = 16K..32K) does not fit the XCD's 4 MB L2, so every round
// streams it again; walking back, a round starts on the tiles the previous round has just left in the L2.  The direction
// depends on the Q block index only (not on what else is in the launch): results do not depend on the batch composition.
#ifndef LBFA_PINGPONG
#define LBFA_PINGPONG 1
#endif
#ifndef LBFA_DMA
#define LBFA_DMA 1  // K / V tiles by LDS-DMA (buffer_load ... lds) instead of staging registers + ds_write: +3..5 %
#endif
#ifndef LBFA_THR
#define LBFA_THR 8.0f
#endif
// fp8 PV: O^T += V^T P^T with ONE v_mfma_scale_f32_32x32x64_f8f6f4 per 32 channels and 64-key tile (e4m3 operands, unit
// block scales: twice the fp16 MFMA rate) - must match the V layout written by lbfa_quant_v_fp8 (quant_kernels.hip)
typedef int i32x8 __attribute__((ext_vector_type(8)));

// l_acc += 1 P^T for one k-step of fp16 P: two v_mfma_f32_4x4x4_16b_f16 with an all-ones A operand (16 independent
// 4x4 blocks: lane l supplies column l & 3 of block l >> 2 and gets that column's sums back, i.e. the sum of its own four
// values, in all four accumulator registers).
__device__ __forceinline__ void rowsum_mfma(f32x4& l_acc, const f16x8& pfrag) {
  const f16x4 ones4 = f16x4{(_Float16)1.0f, (_Float16)1.0f, (_Float16)1.0f, (_Float16)1.0f};
  l_acc = __builtin_amdgcn_mfma_f32_4x4x4f16(ones4, f16x4{pfrag[0], pfrag[1], pfrag[2], pfrag[3]}, l_acc, 0, 0, 0);
  l_acc = __builtin_amdgcn_mfma_f32_4x4x4f16(ones4, f16x4{pfrag[4], pfrag[5], pfrag[6], pfrag[7]}, l_acc, 0, 0, 0);
}

#ifdef LBFA_STAMPS  // diagnostic build only: where does a workgroup's time go (s_memtime at five points, wave 0 lane 0)
__device__ long long g_stamps[8192 * 8];
#define LBFA_STAMP(k)                                                                                             \
  do {                                                                                                            \
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime();     \
  } while (0)
extern "C" int lbfa_debug_stamps(void* dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), sizeof(g_stamps), 0, hipMemcpyDeviceToHost);
}
#else
#define LBFA_STAMP(k)
#endif

template <int D, int QT, int VT, int OT, bool CAUSAL, bool QQ = false>
// D = 64 fits three waves per SIMD (<= 168 registers): ask for it, or an instance one register over silently drops to two
__global__ __launch_bounds__(256, D == 64 ? 3 : 2) void attn_fwd_kernel(AttnParams p) {
  static_assert(!QQ || QT == kQInt8, "in-kernel Q quantisation belongs to the int8 path");
  constexpr bool FP8 = (VT == LBFA_E4M3);
  // QT = kQInt8: the low-bit path.  QT = LBFA_F16 / LBFA_BF16: un-quantised Q and K (the FP16 branch of the
  // precision router, src/core.py:1066-1096): same tiling and softmax, scores from v_mfma_f32_32x32x16_f16 / _bf16 on
  // 16-bit tiles (K rows are 2 D bytes); P and V stay fp16 (bf16 V is converted on the way in, as in the low-bit path).
  constexpr bool QK16 = (QT != kQInt8);
  constexpr int ESZ = QK16 ? 2 : 1;                  // bytes per Q / K element
  constexpr int RB = D * ESZ;                        // bytes per K row
  // fp8 P is scaled so that its maximum is 448 = e4m3 max (attn_utils.cuh:30): no headroom to defer
  constexpr float THR = FP8 ? 0.0f : LBFA_THR;
  constexpr int PRIO = FP8 ? LBFA_PRIO_FP8 : LBFA_PRIO;
  constexpr int KS = RB / 32;                        // k-steps of the score product (32 int8 or 16 fp16 per MFMA)
  constexpr int DB = D / 32;                         // 32-channel blocks of O^T
  constexpr int KBYTES = 64 * RB;                    // K tile
  constexpr int VBYTES = FP8 ? 64 * D : 128 * D;     // V tile
  constexpr int KCH = KBYTES / (256 * 16);           // 16-B chunks per thread
  constexpr int VCH = VBYTES / (256 * 16);
  // K / V tiles go from global memory straight into LDS (`buffer_load_dwordx4 ... lds`): no staging registers, no
  // ds_write pass.  Not for bf16 V, which is converted to fp16 on the way in (registers + ds_write).
  constexpr bool DMA = (LBFA_DMA != 0);                        // K
  constexpr bool DMA_V = (LBFA_DMA != 0) && (VT != LBFA_BF16);  // V
  constexpr int TILES_BYTES = 2 * (KBYTES + VBYTES);
  // ONE LDS object (a second one next to an LDS-DMA target makes hipcc drain vmcnt before every ds_read): the tile
  // buffers + 16 bytes for the workgroup reductions (block amax, overflow vote), which must not alias a tile in flight
  __shared__ __attribute__((aligned(16))) char smem[TILES_BYTES + 16];

  LBFA_STAMP(0);
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

  // ---- extents of this problem: dense batch entry, or sequence b of a packed variable-length batch ----
  int Sq = p.Sq, Sk = p.Sk, nK = p.nK;
  int64_t q_off = (int64_t)b * p.qb, k_off = (int64_t)b * p.kb, v_off = (int64_t)b * p.vb, o_off = (int64_t)b * p.ob;
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
    v_off = (int64_t)k0 * p.vs;
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
  const int dq_valid = QK16 ? p.d_valid : D;  // un-quantised Q / K come from the caller's tensors: head-dim padding applies
  const char* kbase = (const char*)p.k + ESZ * (k_off + (int64_t)hk * p.kh);
  const int64_t k_bytes = ESZ * ((int64_t)(Sk - 1) * p.ks + dq_valid);
  const int64_t k_tile_stride = ESZ * 64 * p.ks;
  const char* vbase;
  int64_t v_bytes, v_tile_stride;  // bytes between consecutive 64-key tiles
  if constexpr (FP8) {
    vbase = (const char*)p.v + (((int64_t)b * p.Hkv + hk) * nK) * (int64_t)(D * 64);
    v_bytes = (int64_t)nK * D * 64;
    v_tile_stride = D * 64;
  } else {
    vbase = (const char*)p.v + 2 * (v_off + (int64_t)hk * p.vh);
    v_bytes = 2 * ((int64_t)(Sk - 1) * p.vs + p.d_valid);
    v_tile_stride = 128 * p.vs;
  }

  // ---- loop-invariant per-thread offsets: global (voffset) and LDS -----------------------------------
  // Chunk i of a thread is chunk 0 moved down by a whole number of rows (KROWS / VROWS per pass), which leaves the
  // swizzles unchanged: one voffset / LDS offset per operand, the rest is a scalar soffset and a ds immediate.
  constexpr int KCPR = RB / 16, KROWS = 256 / KCPR;              // K: 16-B chunks per row, rows per pass
  constexpr int VCPR = FP8 ? 1 : D / 8, VROWS = FP8 ? 0 : 256 / VCPR;
  unsigned k_goff, k_loff, v_goff, v_loff;
  {
    const int row = t / KCPR, ch = t % KCPR;
    // LDS-DMA writes a wave's 64 x 16 bytes linearly (thread t -> byte 16 t of the pass): the swizzle moves to the SOURCE
    // address - the slot (row, ch) of the image holds global chunk ch ^ kx(row)
    const int gch = DMA ? (ch ^ kx<RB>(row)) : ch;
    // padded channels: an offset beyond any window -> the range check returns zeros (windows are < 2 GiB, checked by the C ABI)
    k_goff = gch * 16 < ESZ * dq_valid ? ESZ * (unsigned)row * (unsigned)p.ks + gch * 16 : 0x80000000u;
    k_loff = row * RB + ((ch ^ kx<RB>(row)) << 4);
  }
  if constexpr (FP8) {
    v_goff = t * 16;
    v_loff = 2 * KBYTES + t * 16;
  } else {
    const int row = t / VCPR, ch = t % VCPR;
    const int gch = DMA_V ? ((((ch >> 2) ^ vx<D>(row)) << 2) | (ch & 3)) : ch;
    v_goff = gch * 8 < p.d_valid ? 2 * ((unsigned)row * (unsigned)p.vs) + gch * 16 : 0x80000000u;
    v_loff = 2 * KBYTES + row * (2 * D) + (((ch >> 2) ^ vx<D>(row)) << 6) + ((ch & 3) << 4);
  }
  const unsigned k_gstep = ESZ * KROWS * (unsigned)p.ks;                  // bytes between a thread's K chunks
  const unsigned v_gstep = FP8 ? 4096u : 2u * VROWS * (unsigned)p.vs;
  constexpr int K_LSTEP = KROWS * RB, V_LSTEP = FP8 ? 4096 : VROWS * 2 * D;
  static_assert(K_LSTEP == 4096 && V_LSTEP == 4096, "one pass of 256 threads x 16 bytes");
  // ---- staging registers (register path only) ---------------------------------------------------------
  u32x4 kreg[DMA ? 1 : KCH], vreg[DMA_V ? 1 : VCH];
  // windows are < 2 GiB (checked by the C ABI): remaining bytes in 32-bit scalar arithmetic (the lookahead tile past the
  // end gets a window of 0 bytes)
  const int k_bytes32 = (int)k_bytes, v_bytes32 = (int)v_bytes, k_stride32 = (int)k_tile_stride, v_stride32 = (int)v_tile_stride;
  typedef __attribute__((address_space(3))) void* lds_void_ptr;
  // fetch tile j (into LDS buffer `buf_tag` with DMA, into the staging registers otherwise)
  auto load_tile = [&](int j, auto buf_tag) __attribute__((always_inline)) {  // rows / tiles past the end are outside the descriptor and read as zeros
    constexpr int BUF = decltype(buf_tag)::value;
    const bool in_range = (unsigned)j < (unsigned)nK;  // the look-ahead past either end gets a window of 0 bytes
    const int ko = in_range ? j * k_stride32 : 0, vo = in_range ? j * v_stride32 : 0;
    const int k_rem = in_range ? max(0, k_bytes32 - ko) : 0, v_rem = in_range ? max(0, v_bytes32 - vo) : 0;
    const __amdgpu_buffer_rsrc_t k_rs = make_rsrc(kbase + ko, (unsigned)k_rem);
    const __amdgpu_buffer_rsrc_t v_rs = make_rsrc(vbase + vo, (unsigned)v_rem);
    // DMA destination = wave-uniform base (+ 16 bytes per lane, implicit)
    if constexpr (DMA) {
      char* kdst = smem + BUF * KBYTES + wave * 1024;
#pragma unroll
      for (int i = 0; i < KCH; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rs, (lds_void_ptr)(kdst + i * 4096), 16, (int)k_goff, (int)(i * k_gstep), 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < KCH; ++i) kreg[i] = buf_load16(k_rs, k_goff, i * k_gstep);
    }
    if constexpr (DMA_V) {
      char* vdst = smem + 2 * KBYTES + BUF * VBYTES + wave * 1024;
#pragma unroll
      for (int i = 0; i < VCH; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rs, (lds_void_ptr)(vdst + i * 4096), 16, (int)v_goff, (int)(i * v_gstep), 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < VCH; ++i) vreg[i] = buf_load16(v_rs, v_goff, i * v_gstep);
    }
  };
  auto store_tile = [&](auto buf_tag) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_tag)::value;
    if constexpr (!DMA) {
#pragma unroll
      for (int i = 0; i < KCH; ++i) {
        u32x4 val = kreg[i];
        *reinterpret_cast<u32x4*>(smem + k_loff + i * K_LSTEP + BUF * KBYTES) = val;
      }
    }
    if constexpr (!DMA_V) {
#pragma unroll
      for (int i = 0; i < VCH; ++i) {
        u32x4 val = vreg[i];
        if constexpr (VT == LBFA_BF16) val = bf16x8_to_f16x8(val);  // bf16 -> fp16 on the way in (src/core.py:307-308 `v.to(float16)`)
        *reinterpret_cast<u32x4*>(smem + v_loff + i * V_LSTEP + BUF * VBYTES) = val;
      }
    }
  };

  // processing order of the key tiles: i-th tile processed = tile_of(i)
  constexpr int kRound = (D == 64 && !FP8) ? 96 : 64;  // workgroups an XCD runs at a time (32 CUs x 3 or 2)
  const bool rev = !CAUSAL && (LBFA_PINGPONG != 0) && ((Sk & 63) == 0) && (((qt / kRound) & 1) != 0);  // workgroup-uniform
  auto tile_of = [&](int i) __attribute__((always_inline)) { return rev ? nK - 1 - i : i; };
  load_tile(tile_of(0), std::integral_constant<int, 0>{});  // first K / V tile: in flight while Q is fetched (and quantised)
  // ... and so are the dequantisation scales of the first 64 key tiles (lane l: tile l), needed right after the Q prologue
  const float* ksc = nullptr;
  if constexpr (!QK16) ksc = p.k_scale + ksc_base + (int64_t)hk * p.ksc_h;
  const int ksc_blk = (int)p.ksc_blk;
  float ks_first = 0.f;
  if constexpr (!QK16) ks_first = lane < nK ? ksc[lane * ksc_blk] : 0.f;

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
    const char* qbase = (const char*)p.q + ESZ * (q_off + (int64_t)h * p.qh);
    const __amdgpu_buffer_rsrc_t q_rs = make_rsrc(qbase, (unsigned)(ESZ * ((int64_t)(Sq - 1) * p.qs + dq_valid)));
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const unsigned col_b = 16 * hh + 32 * s;  // byte column
      u32x4 raw = buf_load16(q_rs, col_b < (unsigned)(ESZ * dq_valid) ? ESZ * (unsigned)qrow * (unsigned)p.qs + col_b : 0x80000000u, 0);
      qf[s] = __builtin_bit_cast(i32x4, raw);
    }
    if constexpr (!QK16) qsc = p.q_scale[qsc_base + (int64_t)h * p.qsc_h + (int64_t)qt * p.qsc_blk];
  }
  LBFA_STAMP(1);

  int n_tiles = nK;
  if constexpr (CAUSAL) n_tiles = min(nK, 2 * (qt + 1));

  // Fragment read addresses.  The swizzles depend only on the low row bits, so the block / k-step / high-half /
  // buffer parts are compile-time byte offsets folded into the ds_read immediates; per lane only KS (K) and
  // DB or 4 (V) base registers are needed.
  // K: chunk (2s + hh) ^ kx(r) of row r.  r * RB has no bits below RB, so the address is kf_lane ^ (s << 5) with
  // kf_lane = r * RB + ((hh ^ kx(r)) << 4): with many k-steps (fp16, D = 128) one v_xor per fragment replaces KS registers.
  constexpr bool KF_XOR = (KS >= 8);
  const unsigned kf_lane = r * RB + ((hh ^ kx<RB>(r)) << 4);
  unsigned kf_base[KF_XOR ? 1 : KS];  // + kb2 * 32 * RB
  if constexpr (!KF_XOR) {
#pragma unroll
    for (int s = 0; s < KS; ++s) kf_base[s] = kf_lane ^ (s << 5);
  }
  constexpr int NVB = FP8 ? 1 : DB;
  unsigned vf_base[NVB];  // f16: [db] + ks*16*2D + hi*8*2D ;  fp8: + db*32*64, second 16-B chunk at ^ 16
#pragma unroll
  for (int i = 0; i < NVB; ++i) {
    if constexpr (FP8) {
      // channel row r (+ 32 db) of the [D][64] image, this lane's 32 keys = 16-B chunks 2hh and 2hh+1, chunk c stored at c ^ ((r>>2)&3)
      vf_base[i] = 2 * KBYTES + r * 64 + (((2 * hh) ^ ((r >> 2) & 3)) << 4);
    } else {
      const int vrow = 4 * hh + ((lane & 15) >> 2);
      const int vcol = 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
      vf_base[i] = 2 * KBYTES + vrow * (2 * D) + ((i ^ vx<D>(vrow)) << 6) + vcol;
    }
  }

  // ---- running state --------------------------------------------------------------------------------
  // Row sums.  Each lane sums the keys IT has seen (its half of every tile); both halves of a row share m_run, hence every
  // rescale factor, so the halves are added once, in the epilogue.  MSUM (fp16 P, D = 64): the sum lives on the matrix pipe -
  // l_acc accumulates 1 P^T with an all-ones A operand next to the PV MFMAs (v_mfma_f32_4x4x4_16b_f16: every lane sums
  // its own values), so the 32 v_add_f32 per lane and tile disappear.  Otherwise fp32 adds into l_run (fp8 P: the reference
  // sums before rounding, qk_int_sv_f8_cuda.cu:430-445).  Nothing inside the tile loop reads either sum.
  constexpr bool MSUM = !FP8 && (LBFA_LSUM(D) == 2);
  typedef f32x4 lacc_t;
  f32x16 acc_o[DB];
  lacc_t l_acc;
  float m_run;  // reference max (base-2 domain), identical in both halves of a row
  float l_run;
  auto reset_state = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc_o[db][i] = 0.f;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(lacc_t) / 4); ++i) l_acc[i] = 0.f;
    m_run = -INFINITY;
    l_run = 0.f;
  };
  reset_state();

  // The int8 MFMA accumulates on top of this constant block (kept in registers for the whole kernel).
  i32x16 cmagic;
#pragma unroll
  for (int i = 0; i < 16; ++i) cmagic[i] = kMagicBits;
  asm volatile("" : "+v"(cmagic));  // opaque: otherwise the block is re-materialised from SGPRs with 8 v_mov_b64 in every tile

  // ---- exact bias folding --------------------------------------------------------------------------------
  // With tv = kMagic + s (the accumulator bits) the exponent argument is ONE fma:
  //     s*sc - m  ==  fma(tv, sc, c1),   c1 = -kMagic*sc - m
  // which is exact (one rounding, of the small final result) provided -kMagic*sc and c1 are representable.
  // Both are forced onto a common power-of-two grid G chosen from the largest dequantisation scale of this
  // (batch, kv-head): the per-tile scale is rounded to a multiple of g = G/2^22 (relative change <= 2^-21 *
  // sc_max/sc, far below int8 quantisation noise), so kMagic*sc = 3*k*G exactly, and the softmax reference m
  // - which may be ANY value near the row max - is kept on the same grid (rounded up by < G <= 2^-9 for
  // typical data).  |c1| <= 1.17*kMagic*sc_max < 2^21*G, so every constant is an exact multiple of G.
  float ks_max = 0.f;
  if constexpr (!QK16) {
    ks_max = ks_first;
    for (int i = lane + 64; i < nK; i += 64) ks_max = fmaxf(ks_max, ksc[i * ksc_blk]);
    ks_max = fmaxf(wave_max_nonneg(ks_max), 1e-30f);  // scales are positive (the quantiser floors amax)
  }
  const float sc_max = qsc * ks_max;
  const int gexp = (int)((__float_as_uint(1.25f * kMagic * sc_max) >> 23) & 0xff) - 127 + 1 - 21;  // log2(G)
  const float G = __builtin_ldexpf(1.0f, gexp), invG = __builtin_ldexpf(1.0f, -gexp);
  const float g = __builtin_ldexpf(1.0f, gexp - 22), invg = __builtin_ldexpf(1.0f, 22 - gexp);
  auto grid_up = [&](float m) __attribute__((always_inline)) { return __builtin_ceilf(m * invG) * G; };  // -inf stays -inf
  // Per-tile constants sc (dequantisation scale on the g grid) and c0 = -kMagic * sc are the same for every lane: lane l
  // of the wave computes them for tile 64 c + l once per chunk of 64 tiles, and each tile fetches its pair with two
  // v_readlane (no per-tile global load, no per-tile float math on uniform values).
  float sc_tab = 0.f, c0_tab = 0.f;
  auto refresh_scale_table = [&](int j0) __attribute__((always_inline)) {
    if constexpr (!QK16) {
      const int jt = j0 + lane;
      const float ks_l = j0 == 0 ? ks_first : (jt < nK ? ksc[jt * ksc_blk] : 0.f);
      // at least one grid step: a block whose scale is < 2^-22 of the largest one (an all-zero K block) must not get
      // sc = 0, or a masked key (tv = -inf) would turn into fma(-inf, 0, c1) = NaN
      sc_tab = fmaxf(__builtin_rintf(qsc * ks_l * invg), 1.0f) * g;
      c0_tab = -kMagic * sc_tab;  // exact
    }
  };

  // One 64-key tile.  EXACT = take the exact row max before exponentiating (masked tiles, fp8 P, the exact re-run);
  // otherwise the tile is exponentiated against the reference as it stands (see `lazy softmax reference` below).
  // PRIME = scores + reference only (first tile of the lazy pass).
  auto compute_tile = [&](auto buf_tag, int j, auto masked_tag, auto exact_tag, auto prime_tag) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_tag)::value;
    constexpr bool MASKED = decltype(masked_tag)::value;
    constexpr bool EXACT = decltype(exact_tag)::value || MASKED || FP8;
    constexpr bool PRIME = decltype(prime_tag)::value;
    const char* kbuf = smem + BUF * KBYTES;
    const char* vbuf = smem + BUF * VBYTES;
    // -- online softmax, base 2
    float sc, c0;
    if constexpr (QK16) {
      sc = p.qk_scale;  // fp32 scores: one fma per element, no bias to fold
      c0 = 0.f;
    } else {
      sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc_tab), j & 63));
      c0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c0_tab), j & 63));
    }
    float x[2][16];  // scores as floats kMagic + s (accumulator bits); fp32 P path: overwritten in place by P
    // -- S^T = K Q^T (int8 -> int32, biased by kMagic): one 32-key block
    auto compute_scores = [&](auto kb2_tag) __attribute__((always_inline)) {
      constexpr int kb2 = decltype(kb2_tag)::value;
      i32x16 sacc;
      f32x16 facc;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const unsigned kfa = KF_XOR ? (kf_lane ^ (unsigned)(s << 5)) : kf_base[KF_XOR ? 0 : s];
        const i32x4 kf = *reinterpret_cast<const i32x4*>(kbuf + kfa + kb2 * 32 * RB);
        if constexpr (QT == LBFA_BF16) {  // bf16 Q / K go to the bf16 MFMA as they are: exact products, full bf16 range
          typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
          const bf16x8 ka = __builtin_bit_cast(bf16x8, kf), qb = __builtin_bit_cast(bf16x8, qf[s]);
          if (s == 0) facc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qb, f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0);
          else facc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qb, facc, 0, 0, 0);
        } else if constexpr (QK16) {
          const f16x8 ka = __builtin_bit_cast(f16x8, kf), qb = __builtin_bit_cast(f16x8, qf[s]);
          if (s == 0) facc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka, qb, f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0);
          else facc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka, qb, facc, 0, 0, 0);
        } else {
          if (s == 0) sacc = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf, qf[s], cmagic, 0, 0, 0);
          else sacc = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf, qf[s], sacc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float tv;
        if constexpr (QK16) tv = facc[i];
        else tv = __int_as_float(sacc[i]);
        if constexpr (MASKED) {
          const int key = j * 64 + 32 * kb2 + (i & 3) + 8 * (i >> 2) + 4 * hh;
          bool dead = key >= Sk;
          if constexpr (CAUSAL) dead = dead || (key > qrow);
          if (dead) tv = -INFINITY;  // fma(-inf, sc, c1) = -inf -> p = 0  (sc > 0, see refresh_scale_table)
        }
        x[kb2][i] = tv;
      }
    };
    // Move the reference m_run up to (at least) this tile's row max, rescaling O and l, when some row of the
    // wave needs it.  First tile: m_run = -inf -> alpha = 0.
    auto update_reference = [&](float thr) __attribute__((always_inline)) {
      float tmax = -INFINITY;
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
        for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, x[kb2][i]);
      tmax = half_swap_max(tmax);
      const float xmax = __builtin_fmaf(tmax, sc, c0);  // row max of the dequantised scores; -inf if all masked
      // fp8 P: keep the exact row max as reference so that P_max = 448 = e4m3 max exactly, as the reference
      // specifies (attn_utils.cuh:30); c1 then carries a rounding of <= 2^-13 relative, invisible at 3 mantissa bits.
      const float m_cand = fmaxf(m_run, (FP8 || QK16) ? xmax : grid_up(xmax));
      if (__any(m_cand > m_run + thr)) {
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_cand);  // m_run = -inf -> 0
        m_run = m_cand;
        l_run *= alpha;
        if constexpr (MSUM) {
#pragma unroll
          for (int i = 0; i < (int)(sizeof(lacc_t) / 4); ++i) l_acc[i] *= alpha;
        }
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc_o[db][i] *= alpha;
      }
    };
    typedef typename std::conditional<FP8, long, f16x8>::type pfrag_t;
    pfrag_t pf[4];
    i32x8 pf8;  // fp8: all 32 P values of the lane = one B operand (k = 32 hh + 16 kb2 + i)
    float psum = 0.f;
    // P of one 32-key block: x <- P (fp32 paths), pf / pf8 <- packed P^T fragments (k-steps 2 kb2, 2 kb2 + 1)
    auto exponentiate = [&](auto kb2_tag, float c1) __attribute__((always_inline)) {
      constexpr int kb2 = decltype(kb2_tag)::value;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        x[kb2][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[kb2][i], sc, c1));
        if constexpr (!MSUM) psum += x[kb2][i];
        // keep P in the score registers: left alone, the scheduler issues all exponentials first and sinks the adds, which
        // needs 32 more registers (three waves per SIMD no longer fit at D = 64)
        if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int g8 = 0; g8 < 2; ++g8) {
        const int ks = 2 * kb2 + g8, rb = 8 * g8;
        if constexpr (FP8) {
          unsigned w0 = 0, w1 = 0;
          w0 = __builtin_amdgcn_cvt_pk_fp8_f32(x[kb2][rb + 0], x[kb2][rb + 1], w0, false);
          w0 = __builtin_amdgcn_cvt_pk_fp8_f32(x[kb2][rb + 2], x[kb2][rb + 3], w0, true);
          w1 = __builtin_amdgcn_cvt_pk_fp8_f32(x[kb2][rb + 4], x[kb2][rb + 5], w1, false);
          w1 = __builtin_amdgcn_cvt_pk_fp8_f32(x[kb2][rb + 6], x[kb2][rb + 7], w1, true);
          pf8[2 * ks] = (int)w0;
          pf8[2 * ks + 1] = (int)w1;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) pf[ks][e] = (_Float16)x[kb2][rb + e];
        }
      }
    };
    // O^T += V^T P^T for the two k-steps of one 32-key block (P^T fragments straight from the score accumulators),
    // and the row sums 1 P^T
    auto pv_half = [&](auto kb2_tag) __attribute__((always_inline)) {
      constexpr int kb2 = decltype(kb2_tag)::value;
      if constexpr (!FP8) {
        static_for<0, 2 * DB>([&](auto i) {
          constexpr int idx = decltype(i)::value;
          constexpr int ks = 2 * kb2 + idx / DB, db = idx % DB;
          const f16x4 vlo = lds_read_tr16(vbuf + vf_base[db] + ks * 32 * D);
          const f16x4 vhi = lds_read_tr16(vbuf + vf_base[db] + ks * 32 * D + 16 * D);
          const f16x8 vf = f16x8{vlo[0], vlo[1], vlo[2], vlo[3], vhi[0], vhi[1], vhi[2], vhi[3]};
          acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[ks], acc_o[db], 0, 0, 0);
          if constexpr (MSUM && db == DB - 1) rowsum_mfma(l_acc, pf[ks]);
        });
      }
    };

    compute_scores(std::integral_constant<int, 0>{});
    compute_scores(std::integral_constant<int, 1>{});
    if constexpr (PRIME) {
      update_reference(0.0f);
      return;
    }
    // `lazy softmax reference` (fp16 P, unmasked tiles, first pass): any reference within 2^15 of the row max is as good
    // as the max itself (P is floating point, fp32 accumulate), so the row-max pass (16 v_max3 + a cross-half swap) is
    // skipped and the tile is exponentiated against the reference the first tile set.  A row whose scores outgrow that
    // reference by more than 2^16 overflows fp16 P: the infinity reaches the row's outputs (and l_acc), is seen ONCE after
    // the loop, and the Q block is redone with the exact row max in every tile (`run_tiles`).  No per-tile check.
    if constexpr (EXACT) update_reference(THR);
    float c1 = c0 - m_run;  // exact (grid argument above); +inf while m_run = -inf
    if constexpr (FP8) c1 += kFp8Offset;
    constexpr bool ILV = (LBFA_ILV(D) != 0) && !FP8;
    if constexpr (ILV && LBFA_ILV(D) == 2) {
      // per k-step (16 keys): the PV MFMAs of step r - 1 between the exponentials of step r; only one step's MFMAs are left
      // without VALU cover at the end of the tile
      auto exp_q = [&](auto ks_tag) __attribute__((always_inline)) {
        constexpr int ks = decltype(ks_tag)::value, kb2 = ks >> 1, rb = 8 * (ks & 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          x[kb2][rb + e] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[kb2][rb + e], sc, c1));
          if constexpr (!MSUM) psum += x[kb2][rb + e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[ks][e] = (_Float16)x[kb2][rb + e];
      };
      auto pv_q = [&](auto ks_tag) __attribute__((always_inline)) {
        constexpr int ks = decltype(ks_tag)::value;
        static_for<0, DB>([&](auto i) {
          constexpr int db = decltype(i)::value;
          const f16x4 vlo = lds_read_tr16(vbuf + vf_base[db] + ks * 32 * D);
          const f16x4 vhi = lds_read_tr16(vbuf + vf_base[db] + ks * 32 * D + 16 * D);
          const f16x8 vf = f16x8{vlo[0], vlo[1], vlo[2], vlo[3], vhi[0], vhi[1], vhi[2], vhi[3]};
          acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[ks], acc_o[db], 0, 0, 0);
          if constexpr (MSUM && db == DB - 1) rowsum_mfma(l_acc, pf[ks]);
        });
      };
      exp_q(std::integral_constant<int, 0>{});
      __builtin_amdgcn_sched_barrier(0);
#ifndef LBFA_ILV_PRIO
#define LBFA_ILV_PRIO 2  // s_setprio(1) from here to the end of the tile: +1.6 % at S = 16K (1 = only the last k-step's MFMAs: +0.4 %)
#endif
      if constexpr (LBFA_ILV_PRIO == 2) __builtin_amdgcn_s_setprio(1);
      static_for<1, 4>([&](auto r) {
        constexpr int ks = decltype(r)::value;
        pv_q(std::integral_constant<int, ks - 1>{});
        exp_q(std::integral_constant<int, ks>{});
        constexpr int NM = DB + (MSUM ? 2 : 0);
        constexpr int NV = (MSUM ? 20 : 28) / NM;
        static_for<0, NM>([&](auto) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (LBFA_ILV_PRIO == 1) __builtin_amdgcn_s_setprio(1);
      pv_q(std::integral_constant<int, 3>{});
    } else if constexpr (ILV) {
      exponentiate(std::integral_constant<int, 0>{}, c1);
      // One wave issues in order: back-to-back MFMAs hold its issue slot and overlap nothing of its own.  Pin the PV MFMAs
      // of block 0 BETWEEN the exponentials of block 1 (one MFMA per few VALU instructions: each MFMA runs in the shadow of
      // the VALU work that follows it).
      __builtin_amdgcn_sched_barrier(0);
      pv_half(std::integral_constant<int, 0>{});
      exponentiate(std::integral_constant<int, 1>{}, c1);
      constexpr int NM = 2 * DB + (MSUM ? 4 : 0);  // MFMAs of the block: PV + row sums
      constexpr int NV = (MSUM ? 40 : 56) / NM;    // 16 fma + 16 exp + 8 cvt (+ 16 add) spread over them
      static_for<0, NM>([&](auto) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
      });
      __builtin_amdgcn_sched_barrier(0);
      pv_half(std::integral_constant<int, 1>{});
    } else {
      // one long VALU phase, then one long MFMA phase under s_setprio: at D = 128 measured faster than two half-tile
      // rounds (the other wave of the SIMD exponentiates in the shadow of this wave's MFMAs)
      exponentiate(std::integral_constant<int, 0>{}, c1);
      exponentiate(std::integral_constant<int, 1>{}, c1);
      __builtin_amdgcn_sched_barrier(0);  // phase fence: keeps V fragment reads and row-sum adds where they are (registers)
      if constexpr (PRIO & 2) __builtin_amdgcn_s_setprio(1);
      pv_half(std::integral_constant<int, 0>{});
      pv_half(std::integral_constant<int, 1>{});
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (!MSUM) l_run += psum;
    if constexpr (FP8) {
      static_for<0, DB>([&](auto i) {
        constexpr int db = decltype(i)::value;
        const i32x4 v0 = *reinterpret_cast<const i32x4*>(vbuf + vf_base[0] + db * 2048);
        const i32x4 v1 = *reinterpret_cast<const i32x4*>(vbuf + (vf_base[0] ^ 16u) + db * 2048);
        const i32x8 vf = i32x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        // e4m3 x e4m3 (cbsz = blgp = 0), E8M0 block scales 0x7F = 2^0
        acc_o[db] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, pf8, acc_o[db], 0, 0, 0, 0x7F, 0, 0x7F);
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
    }
    if constexpr (PRIO & 2) __builtin_amdgcn_s_setprio(0);
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
  auto step = [&](auto buf_tag, auto nbuf_tag, int i, auto masked_tag, auto exact_tag) __attribute__((always_inline)) {
    const int j = tile_of(i);
    if (i != 0 && (j & 63) == (rev ? 63 : 0)) refresh_scale_table(j & ~63);  // wave-uniform: entering the next chunk of 64 tiles
    load_tile(tile_of(i + 1), nbuf_tag);
    bool skip = false;
    if constexpr (decltype(masked_tag)::value && CAUSAL) skip = j * 64 > row0 + 31;  // all keys above all rows of this wave
    if (!skip) compute_tile(buf_tag, j, masked_tag, exact_tag, No{});
    store_tile(nbuf_tag);
    __syncthreads();  // with LDS-DMA in flight this waits vmcnt(0) first: tile j + 1 has landed when the barrier opens
  };
  // All tiles of this Q block, in processing order (tile_of).  The first one is in flight or being re-fetched on entry.
  auto run_tiles = [&](auto exact_tag) __attribute__((always_inline)) {
    constexpr bool EX = decltype(exact_tag)::value || FP8;
    refresh_scale_table(tile_of(0) & ~63);
    store_tile(B0{});
    __syncthreads();
    if constexpr (!EX) {
      if (n_main > 0) compute_tile(B0{}, tile_of(0), No{}, No{}, Yes{});  // reference <- exact row max of the first tile
    }
    int j = 0;
    for (; j + 1 < n_main; j += 2) {
      step(B0{}, B1{}, j, No{}, exact_tag);
      step(B1{}, B0{}, j + 1, No{}, exact_tag);
    }
    // here j is even: tile j lives in buffer 0
    for (; j < n_tiles; j += 2) {
      if (j < n_main) step(B0{}, B1{}, j, No{}, exact_tag);
      else step(B0{}, B1{}, j, Yes{}, exact_tag);
      if (j + 1 < n_tiles) {
        if (j + 1 < n_main) step(B1{}, B0{}, j + 1, No{}, exact_tag);
        else step(B1{}, B0{}, j + 1, Yes{}, exact_tag);
      }
    }
  };

  LBFA_STAMP(2);
  run_tiles(No{});
  LBFA_STAMP(3);
  float l_tot;
  auto row_sum = [&]() __attribute__((always_inline)) {
    l_tot = half_swap_sum(MSUM ? l_acc[0] : l_run);
  };
  row_sum();
  if constexpr (!FP8) {
    // Did any P overflow fp16 anywhere in this Q block?  An infinite P makes every output channel of its row +-inf or NaN
    // (inf * 0) and, with row sums on the matrix pipe, the row sum too.  The four waves share the K / V tiles and the
    // barriers, so the decision is taken for the workgroup.  The last step ended with a barrier: smem is free.
    int* flag = reinterpret_cast<int*>(smem + TILES_BYTES);
    float chk = l_tot;
    if constexpr (!MSUM) {
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) chk += fabsf(acc_o[db][i]);
    }
    const int bad = __any(!(chk < INFINITY)) ? 1 : 0;
    if (lane == 0) flag[wave] = bad;
    __syncthreads();
    const int any_bad = flag[0] | flag[1] | flag[2] | flag[3];
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(any_bad)) {
      reset_state();
      load_tile(tile_of(0), B0{});
      run_tiles(Yes{});
      row_sum();
    }
  }

  LBFA_STAMP(4);
  // ---- epilogue: O = O^T / l (x v_scale), LSE ------------------------------------------------------------
  const float inv_l = l_tot > 0.f ? 1.0f / l_tot : 0.f;  // a sequence without keys (packed batches only) yields zeros
  if (qrow < Sq) {
    unsigned short* op = reinterpret_cast<unsigned short*>(p.o) + o_off + (int64_t)h * p.oh + (int64_t)qrow * p.os;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int d0 = 32 * db + 8 * g4 + 4 * hh;
        if (d0 >= p.d_valid) continue;  // d_valid is a multiple of 8
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
      const int64_t li = ((int64_t)b * p.Hq + h) * p.Sq + qrow;  // dense only (the packed entry points take no lse)
      ls *= p.lse_scale;
      if constexpr (QQ) ls += row_corr * p.lse_corr_scale;  // 0 when there is no smoothing vector
      else if (p.lse_corr != nullptr) ls += p.lse_corr[li] * p.lse_corr_scale;
      p.lse[li] = ls;
    }
  }
  LBFA_STAMP(5);
}

// fp16-P variants run on the 16x16 MFMA shapes (attn_fwd16.hip); fp8 PV stays here
#ifndef LBFA_SH16
#define LBFA_SH16 1
#endif
hipError_t launch16_attn_fwd(const AttnParams& p, int D, int v_dtype, int o_dtype, int causal, hipStream_t stream);
hipError_t launch16_attn_fwd_qq(const AttnParams& p, int D, int dtype, int causal, hipStream_t stream);
hipError_t launch16_attn_fwd_f16(const AttnParams& p, int D, int dtype, int causal, hipStream_t stream);

hipError_t launch_attn_fwd(const AttnParams& p, int D, int v_dtype, int o_dtype, int causal, hipStream_t stream) {
  if (LBFA_SH16 && v_dtype != LBFA_E4M3) return launch16_attn_fwd(p, D, v_dtype, o_dtype, causal, stream);
  const unsigned n = (unsigned)p.B * p.Hq * p.nQ;
  dim3 grid(n), block(256);
#define LBFA_A(DD, VT, OT)                                                                                 \
  do {                                                                                                     \
    if (causal) hipLaunchKernelGGL((attn_fwd_kernel<DD, kQInt8, VT, OT, true>), grid, block, 0, stream, p);  \
    else hipLaunchKernelGGL((attn_fwd_kernel<DD, kQInt8, VT, OT, false>), grid, block, 0, stream, p);        \
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

// int8 K codes, Q quantised inside the kernel from its fp16 / bf16 source (dtype = Q's = O's); V of the same dtype or e4m3
hipError_t launch_attn_fwd_qq(const AttnParams& p, int D, int dtype, int v_fp8, int causal, hipStream_t stream) {
  if (LBFA_SH16 && !v_fp8) return launch16_attn_fwd_qq(p, D, dtype, causal, stream);
  const unsigned n = (unsigned)p.B * p.Hq * p.nQ;
  dim3 grid(n), block(256);
#define LBFA_QQ(DD, VT, DT)                                                                                       \
  do {                                                                                                            \
    if (causal) hipLaunchKernelGGL((attn_fwd_kernel<DD, kQInt8, VT, DT, true, true>), grid, block, 0, stream, p);   \
    else hipLaunchKernelGGL((attn_fwd_kernel<DD, kQInt8, VT, DT, false, true>), grid, block, 0, stream, p);         \
  } while (0)
#define LBFA_QQ2(DD)                                                     \
  do {                                                                   \
    if (dtype == LBFA_F16) {                                             \
      if (v_fp8) LBFA_QQ(DD, LBFA_E4M3, LBFA_F16);                       \
      else LBFA_QQ(DD, LBFA_F16, LBFA_F16);                              \
    } else {                                                             \
      if (v_fp8) LBFA_QQ(DD, LBFA_E4M3, LBFA_BF16);                      \
      else LBFA_QQ(DD, LBFA_BF16, LBFA_BF16);                            \
    }                                                                    \
  } while (0)
  if (D == 64) LBFA_QQ2(64);
  else LBFA_QQ2(128);
#undef LBFA_QQ2
#undef LBFA_QQ
  return hipGetLastError();
}

// un-quantised Q / K / V of one dtype (fp16, or bf16 converted to fp16 on the way into LDS / registers)
hipError_t launch_attn_fwd_f16(const AttnParams& p, int D, int dtype, int causal, hipStream_t stream) {
  if (LBFA_SH16) return launch16_attn_fwd_f16(p, D, dtype, causal, stream);
  const unsigned n = (unsigned)p.B * p.Hq * p.nQ;
  dim3 grid(n), block(256);
#define LBFA_F(DD, DT)                                                                                \
  do {                                                                                                \
    if (causal) hipLaunchKernelGGL((attn_fwd_kernel<DD, DT, DT, DT, true>), grid, block, 0, stream, p);  \
    else hipLaunchKernelGGL((attn_fwd_kernel<DD, DT, DT, DT, false>), grid, block, 0, stream, p);        \
  } while (0)
  if (D == 64) { if (dtype == LBFA_F16) LBFA_F(64, LBFA_F16); else LBFA_F(64, LBFA_BF16); }
  else { if (dtype == LBFA_F16) LBFA_F(128, LBFA_F16); else LBFA_F(128, LBFA_BF16); }
#undef LBFA_F
  return hipGetLastError();
}

}  // namespace lbfa
