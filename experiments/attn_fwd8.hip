// Fused low-bit FlashAttention-2 forward for gfx950, FP8 PV: INT8 MFMA for QK^T, block-scaled e4m3 MFMA for PV, on the 16x16
// MFMA shapes with 128-key tiles (v_mfma_i32_16x16x64_i8, v_mfma_scale_f32_16x16x128_f8f6f4).
//
// Replaces the arithmetic of csrc/qattn/qk_int_sv_f8_cuda.cu:46-692 (tile loop :270-363, epilogue :554-579, LSE :689); nothing
// here follows that source's structure.  Same decomposition in the large as attn_fwd16.hip - one workgroup = 4 waves = one
// 128-row Q block (one q_scale), a wave owns 32 query rows as two row blocks of 16, K / V tiles arrive by LDS-DMA into two
// buffers, scores biased by 1.5 * 2^23 so the int32 accumulator bits are floats, dequantisation folded into the exp2 argument -
// but the tile is 128 keys (two k_scale blocks): the block-scaled MFMA of the 16x16 family contracts over 128 values, and the
// 32 scores a lane holds per row block and tile ARE one operand of it.
//
// Lane roles (i = lane & 15, g = lane >> 4):
//   S^T = K Q^T   A = K fragment: key 16 kb + i (kb < 8), row bytes [64 s + 16 g, +16)   (ds_read_b128, serves both row blocks)
//                 B = Q fragment: query 16 rb + i of the wave, same bytes                   (registers, whole kernel)
//                 C: lane holds, for ITS query, the keys 16 kb + 4 g + {0..3}: 2 x 8 x 4 = 64 scores per tile
//   O^T += V^T P^T  one v_mfma_scale_f32_16x16x128_f8f6f4 per 16 channels, row block and tile (e4m3 x e4m3, unit E8M0 block
//                 scales: twice the fp16 rate).  B = the lane's 32 P values of the row block packed to e4m3: byte j = 4 kb + e
//                 of the lane is k = 32 g + j (operand map probed with exact integer data, tools/mfma_probe16.hip) <-> key
//                 16 kb + 4 g + e.  A = V^T fragment with the SAME k order: lbfa_quant_v_fp8 stores each 128-key tile as
//                 [D][128] bytes with key 16 kb + 4 g + e at byte 32 g + 4 kb + e of its channel's row, so a lane reads its 32
//                 bytes with two ds_read_b128 (16-byte chunk c of channel d at c ^ vx8(d): conflict-free, tests/test_lds_swizzle_cpu.py).
//   P is scaled so that its row maximum is 448 = e4m3 max (attn_utils.cuh:30): the exact row max is taken in EVERY tile (no
//   headroom to defer), as lane-partial integer maxima per k_scale block (v_max3_i32 on the accumulator bits), one fma each
//   and two permlane swaps; row sums are fp32 adds of P BEFORE it is rounded to e4m3, as the reference sums
//   (qk_int_sv_f8_cuda.cu:314-317 with CudaCore denominators, attn_utils.cuh:424-445).
#include "attn_common.h"

namespace lbfa {

typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int RB>
__device__ __forceinline__ int kx16(int row);  // K-tile 16-byte chunk swizzle (same image as attn_fwd16.hip)
template <> __device__ __forceinline__ int kx16<64>(int row) { return (row >> 1) & 3; }
template <> __device__ __forceinline__ int kx16<128>(int row) { return row & 7; }
// V-tile 16-byte chunk swizzle of channel row d (rows of 128 bytes = 8 chunks); must match vfp8 layout in quant_kernels.hip
__device__ __forceinline__ int vx8(int d) { return (((d >> 3) & 1) << 2) | ((d >> 1) & 1); }

#ifndef LBFA_F8_NRB
#define LBFA_F8_NRB 1  // 16-row blocks per wave of the fp8 kernel (see the template)
#endif
#ifndef LBFA_MX_NOP
#define LBFA_MX_NOP 7  // wait states added behind the block-scaled MFMAs of a tile (tools/check_mfma_hazards.py explains the count)
#endif

// OT = dtype of O (and of the Q source when QQ): int8 K codes (and Q codes unless QQ), e4m3 V.
// NRB = 16-row blocks per wave: a workgroup is 8 / NRB waves.  NRB = 1 (512 threads, 128 VGPRs, 4 waves per SIMD) trades twice the K / V
// fragment reads per row for twice the waves to hide latencies with; NRB = 2 is the 4-wave decomposition of attn_fwd16.hip.
template <int D, int OT, bool CAUSAL, bool QQ, int NRB>
__global__ __launch_bounds__(512 / NRB, NRB == 1 ? 4 : (D == 64 ? 3 : 2)) void attn_fwd8_kernel(AttnParams p) {
  constexpr int NT = 512 / NRB;            // threads
  constexpr int NW = NT / 64;              // waves
  constexpr int RB = D;                    // bytes per K row
  constexpr int KS = RB / 64;              // k-steps of the score product
  constexpr int CB = D / 16;               // 16-channel blocks of O^T
  constexpr int TK = 128;                  // keys per tile
  constexpr int KBYTES = TK * RB, VBYTES = TK * D;
  constexpr int PASS = NT * 16;            // bytes one pass of the workgroup moves
  constexpr int KCH = KBYTES / PASS, VCH = VBYTES / PASS;  // 16-byte chunks per thread
  constexpr int TILES_BYTES = 2 * (KBYTES + VBYTES);
  __shared__ __attribute__((aligned(16))) char smem[TILES_BYTES + 32];  // ONE LDS object (tiles + workgroup reduction)

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

  const int Sq = p.Sq, Sk = p.Sk;
  const int nB = p.nK;               // 64-key scale blocks
  const int nT = (Sk + TK - 1) / TK;  // 128-key tiles
  const int64_t q_off = (int64_t)b * p.qb, k_off = (int64_t)b * p.kb, o_off = (int64_t)b * p.ob;
  const int64_t qsc_base = (int64_t)b * p.qsc_b, ksc_base = (int64_t)b * p.ksc_b;
  const int row0 = qt * 128 + wave * 16 * NRB;  // first query row of this wave
  auto qrow_of = [&](int rb) __attribute__((always_inline)) { return row0 + 16 * rb + i16; };

  // ---- Q rows of this wave: requested first (see attn_fwd16.hip) ---------------------------------------------------
  constexpr int QPC = QQ ? 2 : 1;
  u32x4 qraw[NRB][KS][QPC];
  {
    constexpr int QESZ = QQ ? 2 : 1;
    const int q_valid = QQ ? p.d_valid : D;
    const char* qbase = (const char*)p.q + QESZ * (q_off + (int64_t)h * p.qh);
    const __amdgpu_buffer_rsrc_t q_rs = make_rsrc(qbase, (unsigned)(QESZ * ((int64_t)(Sq - 1) * p.qs + q_valid)));
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int hf = 0; hf < QPC; ++hf) {
          const unsigned col_b = QQ ? 2 * (64 * s + 16 * g + 8 * hf) : 64 * s + 16 * g;  // byte column
          qraw[rb][s][hf] = buf_load16(q_rs, col_b < (unsigned)(QESZ * q_valid) ? QESZ * (unsigned)qrow_of(rb) * (unsigned)p.qs + col_b : 0x80000000u, 0);
        }
  }

  // ---- operand windows (bytes) and the tile fetch ----------------------------------------------------------------------
  const char* kbase = (const char*)p.k + (k_off + (int64_t)hk * p.kh);
  const int k_bytes32 = (int)((int64_t)(Sk - 1) * p.ks + D);
  const int k_stride32 = (int)(TK * p.ks);
  const char* vbase = (const char*)p.v + (((int64_t)b * p.Hkv + hk) * nT) * (int64_t)VBYTES;
  const int v_bytes32 = nT * VBYTES;
  constexpr int KCPR = RB / 16, KROWS = NT / KCPR;
  unsigned k_goff;
  {
    const int row = t / KCPR, ch = t % KCPR;
    // LDS-DMA writes linearly: the slot (row, ch) of the image holds global chunk ch ^ kx16(row)
    k_goff = (unsigned)row * (unsigned)p.ks + ((ch ^ kx16<RB>(row)) << 4);
  }
  const unsigned v_goff = t * 16;  // the V image is copied as it lies in HBM
  const unsigned k_gstep = KROWS * (unsigned)p.ks;
  static_assert(KROWS * RB == PASS, "one pass of the workgroup x 16 bytes");
  typedef __attribute__((address_space(3))) void* lds_void_ptr;
  auto load_tile = [&](int j, auto buf_tag) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_tag)::value;
    const bool in_range = (unsigned)j < (unsigned)nT;  // the look-ahead past either end gets a window of 0 bytes
    const int ko = in_range ? j * k_stride32 : 0, vo = in_range ? j * VBYTES : 0;
    const int k_rem = in_range ? max(0, k_bytes32 - ko) : 0, v_rem = in_range ? max(0, v_bytes32 - vo) : 0;
    const __amdgpu_buffer_rsrc_t k_rs = make_rsrc(kbase + ko, (unsigned)k_rem);
    const __amdgpu_buffer_rsrc_t v_rs = make_rsrc(vbase + vo, (unsigned)v_rem);
    char* kdst = smem + BUF * KBYTES + wave * 1024;  // DMA destination: wave-uniform base (+ 16 bytes per lane, implicit)
#pragma unroll
    for (int c = 0; c < KCH; ++c)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rs, (lds_void_ptr)(kdst + c * PASS), 16, (int)k_goff, (int)(c * k_gstep), 0, 0);
    char* vdst = smem + 2 * KBYTES + BUF * VBYTES + wave * 1024;
#pragma unroll
    for (int c = 0; c < VCH; ++c)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rs, (lds_void_ptr)(vdst + c * PASS), 16, (int)v_goff, (int)(c * PASS), 0, 0);
  };

  // processing order of the key tiles (ping-pong per round of Q blocks: every other round walks the tiles backwards and starts
  // on what the previous round left in the XCD's L2; the direction depends on the Q block index only)
  constexpr int kRound = (D == 64) ? 96 : 64;
  const bool rev = !CAUSAL && ((Sk & (TK - 1)) == 0) && (((qt / kRound) & 1) != 0);
  auto tile_of = [&](int i) __attribute__((always_inline)) { return rev ? nT - 1 - i : i; };
  load_tile(tile_of(0), std::integral_constant<int, 0>{});
  const float* ksc = p.k_scale + ksc_base + (int64_t)hk * p.ksc_h;
  const int ksc_blk = (int)p.ksc_blk;
  // dequantisation scales of the 64 scale blocks around the first tile (lane l: block 64 c + l)
  const int blk_first = (2 * tile_of(0)) & ~63;
  const float ks_first = blk_first + lane < nB ? ksc[(blk_first + lane) * ksc_blk] : 0.f;

  // ---- Q fragments (B operand of the score MFMAs) ------------------------------------------------------------------
  i32x4 qf[NRB][KS];
  float qsc = 1.0f;
  float row_corr[NRB] = {};
  if constexpr (QQ) {
    // in-kernel Q quantiser: same arithmetic as quant_per_block_kernel (src/triton/quant_per_block.py:132-178)
    float xs[NRB][KS][16];
    float amax = 0.f;
    const unsigned short* vec = p.q_dot_vec ? p.q_dot_vec + ((int64_t)b * p.Hkv + hk) * D : nullptr;
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
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
    float* red = reinterpret_cast<float*>(smem + TILES_BYTES);
    if (lane == 0) red[wave] = amax;
    __syncthreads();
    amax = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) amax = fmaxf(amax, red[w]);
    const float scale = fmaxf(amax, 1e-7f) / p.q_qmax;
    qsc = scale;
    const float rcp = 1.0f / scale;
    const bool exact_rcp_ok = (__builtin_amdgcn_readfirstlane(__float_as_uint(scale)) & 0x7fffffu) != 0x7fffffu;
    auto encode = [&](auto fast_tag) __attribute__((always_inline)) {
      constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb)
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
          qf[rb][s] = i32x4{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
        }
    };
    if (exact_rcp_ok) encode(std::true_type{});
    else encode(std::false_type{});
  } else {
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int s = 0; s < KS; ++s) qf[rb][s] = __builtin_bit_cast(i32x4, qraw[rb][s][0]);
    qsc = p.q_scale[qsc_base + (int64_t)h * p.qsc_h + (int64_t)qt * p.qsc_blk];
  }

  int n_tiles = nT;
  if constexpr (CAUSAL) n_tiles = min(nT, qt + 1);

  // ---- fragment read addresses (lane parts; block / k-step / buffer parts are immediates) ------------------------
  const unsigned kf_lane = i16 * RB + ((g ^ kx16<RB>(i16)) << 4);                       // k-step s: ^ (s << 6); key block: + kb * 16 * RB
  const unsigned vf_lane = 2 * KBYTES + i16 * 128 + (((2 * g) ^ vx8(i16)) << 4);        // second chunk: ^ 16; channel block: + cb * 2048

  // ---- running state --------------------------------------------------------------------------------
  f32x4 acc_o[NRB][CB];
  float m_run[NRB], l_run[NRB];
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc_o[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    m_run[rb] = -INFINITY;
    l_run[rb] = 0.f;  // this lane's share of the row sum (its keys of every tile); the four shares meet in the epilogue
  }
  i32x4 cmagic = i32x4{kMagicBits, kMagicBits, kMagicBits, kMagicBits};
  asm volatile("" : "+v"(cmagic));  // opaque: otherwise re-materialised in every tile

  // ---- bias folding: s * sc - m + 8.807 == fma(tv, sc, c1), tv = kMagic + s (the accumulator bits), c1 = -kMagic * sc - m + 8.807.
  // The per-block scale is rounded to a multiple of gg = G / 2^22, G a power of two chosen from the largest dequantisation scale of
  // this (batch, kv-head), so kMagic * sc is exact; m is the exact row max and c1 carries a rounding of <= 2^-13 relative,
  // invisible at 3 mantissa bits.
  float ks_max = 0.f;
  for (int i = lane; i < nB; i += 64) ks_max = fmaxf(ks_max, ksc[i * ksc_blk]);
  ks_max = fmaxf(wave_max_nonneg(ks_max), 1e-30f);
  auto uniform = [](float v) __attribute__((always_inline)) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
  };
  qsc = uniform(qsc);
  const float sc_max = qsc * ks_max;
  const int gexp = (int)((__float_as_uint(1.25f * kMagic * sc_max) >> 23) & 0xff) - 127 + 1 - 21;  // log2(G)
  const float gg = uniform(__builtin_ldexpf(1.0f, gexp - 22)), invg = uniform(__builtin_ldexpf(1.0f, 22 - gexp));
  // Wide scores (one integer step of a score >= 1/8 binade, e.g. randint(-100, 100) inputs): |c0| = kMagic * sc grows past 2^20 and
  // the rounding of c1 = c0 - m + 8.807 (half an ulp of c0) past 2^-4 binades; the bias then comes off the scores with one exact
  // subtraction each (tv - kMagic = s) and the scales stay unrounded: 64 more VALU per tile, for such inputs only.
  const bool wide = gexp >= 0;
  float sc_tab = 0.f, c0_tab = 0.f;
  auto refresh_scale_table = [&](int blk0) __attribute__((always_inline)) {  // blk0: multiple of 64
    const int bl = blk0 + lane;
    const float ks_l = blk0 == blk_first ? ks_first : (bl < nB ? ksc[bl * ksc_blk] : 0.f);
    // at least one grid step: a block whose scale is < 2^-22 of the largest (an all-zero K block) must not get sc = 0, or a
    // masked key (tv = -inf) would turn into fma(-inf, 0, c1) = NaN
    sc_tab = fmaxf(__builtin_rintf(qsc * ks_l * invg), 1.0f) * gg;
    c0_tab = -kMagic * sc_tab;  // exact
    if (wide) {
      sc_tab = fmaxf(qsc * ks_l, 1e-30f);
      c0_tab = 0.f;
    }
  };

  // One 128-key tile: S^T of every row block (a K fragment serves them all), then per row block the online softmax
  // (branch-free reference update; the accumulator rescale it may call for is the one branch), then its PV product.
  auto compute_tile = [&](auto buf_tag, int j, auto masked_tag, auto wide_tag) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_tag)::value;
    constexpr bool MASKED = decltype(masked_tag)::value;
    constexpr bool WIDE = decltype(wide_tag)::value;  // scores as the integers themselves (see `wide` above)
    const char* kbuf = smem + BUF * KBYTES;
    const char* vbuf = smem + BUF * VBYTES;
    float sc[2], c0[2];  // the tile's two k_scale blocks
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      sc[hb] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc_tab), (2 * j + hb) & 63));
      c0[hb] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c0_tab), (2 * j + hb) & 63));
    }
    float x[NRB][8][4];  // [row block][key block][key 4 g + e]: kMagic + s (accumulator bits), then P in place
    i32x8 pf8[NRB];      // packed P^T operands: byte 4 kb + e of the lane = key 16 kb + 4 g + e
    int live[NRB] = {};  // masked tiles: the lane's keys 16 kb + e (+ 4 g) of this tile are alive below this bound
    if constexpr (MASKED) {
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        live[rb] = (CAUSAL ? min(Sk, qrow_of(rb) + 1) : Sk) - j * TK - 4 * g;
        asm volatile("" : "+v"(live[rb]));  // one compare + select per score, not 32 lane masks parked in SGPRs
      }
    }
    // -- S^T = K Q^T (int8 -> int32, biased by kMagic); K fragments of key block kb + 1 requested before the MFMAs of block kb
    i32x4 kfr[2][KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) kfr[0][s] = *reinterpret_cast<const i32x4*>(kbuf + (kf_lane ^ (unsigned)(s << 6)));
    static_for<0, 8>([&](auto kb_tag) {
      constexpr int kb = decltype(kb_tag)::value;
      if constexpr (kb + 1 < 8) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
          kfr[(kb + 1) & 1][s] = *reinterpret_cast<const i32x4*>(kbuf + (kf_lane ^ (unsigned)(s << 6)) + (kb + 1) * 16 * RB);
      }
      i32x4 sacc[NRB];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
          if (s == 0) sacc[rb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(kfr[kb & 1][s], qf[rb][s], cmagic, 0, 0, 0);
          else sacc[rb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(kfr[kb & 1][s], qf[rb][s], sacc[rb], 0, 0, 0);
        }
      }
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float tv = __int_as_float(sacc[rb][e]);
          if constexpr (MASKED) {
            if (16 * kb + e >= live[rb]) tv = -INFINITY;  // fma(-inf, sc, c1) = -inf -> p = 0  (sc > 0, see refresh_scale_table)
          }
          x[rb][kb][e] = tv;
        }
      __builtin_amdgcn_sched_barrier(0);
    });
    // -- online softmax of one row block, base 2: m_run up to this tile's row max, l rescaled, P with its fp32 row sums (before
    // rounding) and the packed operand.  Returns alpha: what the accumulators still have to be multiplied by.
    auto softmax = [&](int rb) __attribute__((always_inline)) {
      float xm[2];
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {  // lane-partial maximum per scale block, as an order key (attn_common.h)
        float m = key_max3<true>(x[rb][4 * hb][0], x[rb][4 * hb][1], x[rb][4 * hb][2]);
        m = key_max3<true>(m, x[rb][4 * hb][3], x[rb][4 * hb + 1][0]);
        m = key_max3<true>(m, x[rb][4 * hb + 1][1], x[rb][4 * hb + 1][2]);
        m = key_max3<true>(m, x[rb][4 * hb + 1][3], x[rb][4 * hb + 2][0]);
        m = key_max3<true>(m, x[rb][4 * hb + 2][1], x[rb][4 * hb + 2][2]);
        m = key_max3<true>(m, x[rb][4 * hb + 2][3], x[rb][4 * hb + 3][0]);
        m = key_max3<true>(m, x[rb][4 * hb + 3][1], x[rb][4 * hb + 3][2]);
        m = key_max<true>(m, x[rb][4 * hb + 3][3]);
        xm[hb] = __builtin_fmaf(WIDE ? m - kMagic : m, sc[hb], c0[hb]);  // dequantised; -inf if all masked
      }
      const float xmax = rows4_key_max<false>(fmaxf(xm[0], xm[1]));
      const float m_new = fmaxf(m_run[rb], xmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run[rb] - m_new);  // m_run = -inf -> 0; unchanged -> 1
      m_run[rb] = m_new;
      const float c1a = c0[0] - m_new + kFp8Offset, c1b = c0[1] - m_new + kFp8Offset;
      if constexpr (WIDE) {
#pragma unroll
        for (int kb = 0; kb < 8; ++kb)
#pragma unroll
          for (int e = 0; e < 4; ++e) x[rb][kb][e] -= kMagic;
      }
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          x[rb][kb][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[rb][kb][e], sc[kb >> 2], kb < 4 ? c1a : c1b));
          psum += x[rb][kb][e];
        }
        unsigned w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(x[rb][kb][0], x[rb][kb][1], w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(x[rb][kb][2], x[rb][kb][3], w, true);
        pf8[rb][kb] = (int)w;
      }
      l_run[rb] = l_run[rb] * alpha + psum;
      return alpha;
    };
    float alpha[NRB];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) alpha[rb] = softmax(rb);
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      if (__any(alpha[rb] != 1.0f)) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc_o[rb][cb][e] *= alpha[rb];
      }
    }
    // -- O^T += V^T P^T: e4m3 x e4m3 (cbsz = blgp = 0), E8M0 block scales 0x7F = 2^0; a V^T fragment serves every row block
    // (fragment of block cb + 1 requested before the MFMAs of block cb; the fences keep the compiler from hoisting all 2 CB reads
    // to the top of the phase, which the 128-register budget of the 8-wave form cannot hold)
    i32x4 va[2], vb[2];
    va[0] = *reinterpret_cast<const i32x4*>(vbuf + vf_lane);
    vb[0] = *reinterpret_cast<const i32x4*>(vbuf + (vf_lane ^ 16u));
    static_for<0, CB>([&](auto c) {
      constexpr int cb = decltype(c)::value;
      if constexpr (cb + 1 < CB) {
        va[(cb + 1) & 1] = *reinterpret_cast<const i32x4*>(vbuf + vf_lane + (cb + 1) * 2048);
        vb[(cb + 1) & 1] = *reinterpret_cast<const i32x4*>(vbuf + (vf_lane ^ 16u) + (cb + 1) * 2048);
      }
      const i32x4 v0 = va[cb & 1], v1 = vb[cb & 1];
      const i32x8 vf = i32x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb)
        acc_o[rb][cb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(vf, pf8[rb], acc_o[rb][cb], 0, 0, 0, 0x7F, 0, 0x7F);
      __builtin_amdgcn_sched_barrier(0);
    });
#if LBFA_MX_NOP >= 0  // (-1: build without the pad, for tools/check_mfma_hazards.py to show what the compiler leaves)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop %0" ::"n"(LBFA_MX_NOP));
    __builtin_amdgcn_sched_barrier(0);
#endif
  };

  // ---- tile loop: one barrier per tile, buffers alternate statically (loop unrolled by two).  Full tiles first, branch-free;
  // then the masked tile (causal diagonal = the Q block's own 128 keys, or a ragged last tile).
  int n_main = n_tiles;
  if constexpr (CAUSAL) n_main = min(n_tiles, qt);
  else if ((Sk & (TK - 1)) != 0) n_main = n_tiles - 1;
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  using No = std::false_type;
  using Yes = std::true_type;
  auto step = [&](auto buf_tag, auto nbuf_tag, int i, auto masked_tag) __attribute__((always_inline)) {
    const int j = tile_of(i);
    if (i != 0 && ((2 * j + (rev ? 1 : 0)) & 63) == (rev ? 63 : 0)) refresh_scale_table((2 * j) & ~63);
    load_tile(tile_of(i + 1), nbuf_tag);
    if (wide) compute_tile(buf_tag, j, masked_tag, Yes{});  // wave-uniform
    else compute_tile(buf_tag, j, masked_tag, No{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next tile has landed (this wave's share) when the barrier opens
    __syncthreads();
  };
  refresh_scale_table(blk_first);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  {
    int i = 0;
    for (; i + 1 < n_main; i += 2) {
      step(B0{}, B1{}, i, No{});
      step(B1{}, B0{}, i + 1, No{});
    }
    for (; i < n_tiles; i += 2) {
      if (i < n_main) step(B0{}, B1{}, i, No{});
      else step(B0{}, B1{}, i, Yes{});
      if (i + 1 < n_tiles) {
        if (i + 1 < n_main) step(B1{}, B0{}, i + 1, No{});
        else step(B1{}, B0{}, i + 1, Yes{});
      }
    }
  }

  // ---- epilogue: O = O^T / l x v_scale, LSE ------------------------------------------------------------
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb) {
    const int qrow = qrow_of(rb);
    const float l_tot = rows4_sum(l_run[rb]);
    const float inv_l = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (qrow < Sq) {
      unsigned short* op = reinterpret_cast<unsigned short*>(p.o) + o_off + (int64_t)h * p.oh + (int64_t)qrow * p.os;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const int d0 = 16 * cb + 4 * g;
        if (d0 >= p.d_valid) continue;  // d_valid is a multiple of 8
        const f32x4 vs4 = *reinterpret_cast<const f32x4*>(p.v_scale + ((int64_t)b * p.Hkv + hk) * D + d0);
        float o4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = acc_o[rb][cb][e] * inv_l * vs4[e];
        uint2 pk;
        pk.x = (unsigned)store_cvt<OT>(o4[0]) | ((unsigned)store_cvt<OT>(o4[1]) << 16);
        pk.y = (unsigned)store_cvt<OT>(o4[2]) | ((unsigned)store_cvt<OT>(o4[3]) << 16);
        *reinterpret_cast<uint2*>(op + d0) = pk;
      }
      if (p.lse != nullptr && g == 0) {
        float ls = log2f(l_tot) + m_run[rb] - kFp8Offset;  // base-2 domain (attn_qk_int8_per_block.py:164-167, qk_int_sv_f8_cuda.cu:689)
        const int64_t li = ((int64_t)b * p.Hq + h) * p.Sq + qrow;
        ls *= p.lse_scale;
        if constexpr (QQ) ls += row_corr[rb] * p.lse_corr_scale;
        else if (p.lse_corr != nullptr) ls += p.lse_corr[li] * p.lse_corr_scale;
        p.lse[li] = ls;
      }
    }
  }
}

// ---- launchers: every fp16-P variant runs in attn_fwd16.hip, fp8 PV here ------------------------------------------
hipError_t launch16_attn_fwd_d64(const AttnParams& p, int v_dtype, int o_dtype, int causal, hipStream_t stream);
hipError_t launch16_attn_fwd_d128(const AttnParams& p, int v_dtype, int o_dtype, int causal, hipStream_t stream);
hipError_t launch16_attn_fwd_qq_d64(const AttnParams& p, int dtype, int causal, hipStream_t stream);
hipError_t launch16_attn_fwd_qq_d128(const AttnParams& p, int dtype, int causal, hipStream_t stream);
hipError_t launch16_attn_fwd_f16_d64(const AttnParams& p, int dtype, int causal, hipStream_t stream);
hipError_t launch16_attn_fwd_f16_d128(const AttnParams& p, int dtype, int causal, hipStream_t stream);

#define LBFA_LAUNCH_FP8(DD, OT, QQ)                                                                                  \
  do {                                                                                                               \
    if (causal) hipLaunchKernelGGL((attn_fwd8_kernel<DD, OT, true, QQ, LBFA_F8_NRB>), grid, block, 0, stream, p);    \
    else hipLaunchKernelGGL((attn_fwd8_kernel<DD, OT, false, QQ, LBFA_F8_NRB>), grid, block, 0, stream, p);          \
  } while (0)

// int8 Q / K codes; V fp16 / bf16 / e4m3 ([D][128]-per-tile image of lbfa_quant_v_fp8)
hipError_t launch_attn_fwd(const AttnParams& p, int D, int v_dtype, int o_dtype, int causal, hipStream_t stream) {
  if (v_dtype != LBFA_E4M3)
    return D == 64 ? launch16_attn_fwd_d64(p, v_dtype, o_dtype, causal, stream) : launch16_attn_fwd_d128(p, v_dtype, o_dtype, causal, stream);
  dim3 grid((unsigned)p.B * p.Hq * p.nQ), block(512 / LBFA_F8_NRB);
  if (D == 64) { if (o_dtype == LBFA_F16) LBFA_LAUNCH_FP8(64, LBFA_F16, false); else LBFA_LAUNCH_FP8(64, LBFA_BF16, false); }
  else { if (o_dtype == LBFA_F16) LBFA_LAUNCH_FP8(128, LBFA_F16, false); else LBFA_LAUNCH_FP8(128, LBFA_BF16, false); }
  return hipGetLastError();
}

// int8 K codes, Q quantised inside the kernel from its fp16 / bf16 source (dtype = Q's = O's); V of the same dtype or e4m3
hipError_t launch_attn_fwd_qq(const AttnParams& p, int D, int dtype, int v_fp8, int causal, hipStream_t stream) {
  if (!v_fp8) return D == 64 ? launch16_attn_fwd_qq_d64(p, dtype, causal, stream) : launch16_attn_fwd_qq_d128(p, dtype, causal, stream);
  dim3 grid((unsigned)p.B * p.Hq * p.nQ), block(512 / LBFA_F8_NRB);
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
