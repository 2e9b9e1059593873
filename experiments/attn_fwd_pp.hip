// 8-wave "ping-pong" variant of the fused low-bit attention forward (gfx950), for long sequences.
//
// Why: per 64x32 score tile one wave needs ~700 cycles of softmax VALU work (v_exp_f32 alone is 8 cycles per
// wave-instruction on gfx950) and 384 (D=64) / 768 (D=128) cycles of MFMA.  Independent waves drift into the
// same phase and the two pipes end up mostly serialised (measured: VALU||MFMA co-execution ~50 % of MFMA
// time in attn_fwd.hip).  Here the two waves that share a SIMD are forced into complementary phases:
//
//   workgroup = 8 waves = 256 query rows; waves w and w+4 share a SIMD.  Group g = w>>2 owns q-block g of
//   the pair.  Time is cut into steps separated by ONE workgroup barrier each; in every step one group runs
//   its MFMA segment  { PV(t-1) ; QK(t) }  while the other runs its VALU segment  { softmax(t') ; pack P }:
//
//        step     0        1        2        3        4    ...
//        group0   QK0      SM0      PV0 QK1  SM1      PV1 QK2
//        group1   -        QK0      SM0      PV0 QK1  SM1
//
//   K/V tiles are shared by all 8 waves through a 3-deep LDS ring: tile t is read in steps 2t .. 2t+3 and
//   tile t+3 is written into its slot at the start of step 2t+5; the global fetch of a tile is issued one
//   tile period before it is written (16-byte buffer loads, staged in registers).
//
// The arithmetic is exactly that of attn_fwd.hip (score-bias trick, exact single-fma exponent on a common
// grid, lazy softmax reference, -inf masking); see that file for the derivations.
#include "attn_common.h"

namespace lbfa {

#ifndef LBFA_PP_PRIO
#define LBFA_PP_PRIO 0
#endif

template <int D, int VT, int OT, bool CAUSAL>
__global__ __launch_bounds__(512, 2) void attn_fwd_pp_kernel(AttnParams p) {
  constexpr bool FP8 = (VT == LBFA_E4M3);
  constexpr int KS = D / 32;
  constexpr int DB = D / 32;
  constexpr int KBYTES = 64 * D;
  constexpr int VBYTES = FP8 ? 64 * D : 128 * D;
  constexpr int NT = 512;
  constexpr int KCH = (KBYTES / 16 + NT - 1) / NT;  // 16-B chunks per thread (the last one may be partial)
  constexpr int VCH = (VBYTES / 16 + NT - 1) / NT;
  constexpr int NBUF = 3;
  constexpr int VBASE = NBUF * KBYTES;  // LDS: [K ring][V ring]
  __shared__ __attribute__((aligned(16))) char smem[NBUF * (KBYTES + VBYTES)];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = wave >> 2;
  const int r = lane & 31, hh = lane >> 5;

  const int nQ2 = (p.Sq + 255) / 256;
  const unsigned w_id = xcd_remap(blockIdx.x, gridDim.x);
  int qt2 = (int)(w_id % (unsigned)nQ2);
  const int bh = (int)(w_id / (unsigned)nQ2);
  if constexpr (CAUSAL) qt2 = nQ2 - 1 - qt2;  // heaviest blocks of a head first
  const int b = bh / p.Hq, h = bh % p.Hq, hk = h / p.group;

  const int row0 = qt2 * 256 + wave * 32;  // first query row of this wave
  const int qrow = row0 + r;

  // ---- operand windows (bytes); descriptors re-based per tile with scalar arithmetic ---------------------
  const char* qbase = (const char*)p.q + (int64_t)b * p.qb + (int64_t)h * p.qh;
  const char* kbase = (const char*)p.k + (int64_t)b * p.kb + (int64_t)hk * p.kh;
  const int64_t k_bytes = (int64_t)(p.Sk - 1) * p.ks + D;
  const int64_t k_tile_stride = 64 * p.ks;
  const char* vbase;
  int64_t v_bytes, v_tile_stride;
  if constexpr (FP8) {
    vbase = (const char*)p.v + (((int64_t)b * p.Hkv + hk) * p.nK) * (int64_t)(D * 64);
    v_bytes = (int64_t)p.nK * D * 64;
    v_tile_stride = D * 64;
  } else {
    vbase = (const char*)p.v + 2 * ((int64_t)b * p.vb + (int64_t)hk * p.vh);
    v_bytes = 2 * ((int64_t)(p.Sk - 1) * p.vs + D);
    v_tile_stride = 128 * p.vs;
  }
  const __amdgpu_buffer_rsrc_t q_rs = make_rsrc(qbase, (unsigned)((int64_t)(p.Sq - 1) * p.qs + D));

  i32x4 qf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s)
    qf[s] = __builtin_bit_cast(i32x4, buf_load16(q_rs, (unsigned)qrow * (unsigned)p.qs + 16 * hh + 32 * s, 0));
  const int qblk = min(2 * qt2 + grp, p.nQ - 1);  // a block past Sq only exists as padding rows
  const float qsc = p.q_scale[((int64_t)b * p.Hq + h) * p.nQ + qblk];
  const float* ksc = p.k_scale + ((int64_t)b * p.Hkv + hk) * p.nK;

  int n_tiles = p.nK;
  if constexpr (CAUSAL) n_tiles = min(p.nK, 4 * (qt2 + 1));
  const bool ragged = (p.Sk & 63) != 0;

  // ---- loop-invariant per-thread offsets -----------------------------------------------------------------
  unsigned k_goff[KCH], k_loff[KCH], v_goff[VCH], v_loff[VCH];
#pragma unroll
  for (int i = 0; i < KCH; ++i) {
    const int c = t + NT * i, row = c / (D / 16), ch = c % (D / 16);
    k_goff[i] = (unsigned)row * (unsigned)p.ks + ch * 16;
    k_loff[i] = row * D + ((ch ^ kx<D>(row)) << 4);
  }
#pragma unroll
  for (int i = 0; i < VCH; ++i) {
    if constexpr (FP8) {
      v_goff[i] = (t + NT * i) * 16;
      v_loff[i] = VBASE + (t + NT * i) * 16;
    } else {
      const int c = t + NT * i, row = c / (D / 8), ch = c % (D / 8);
      v_goff[i] = 2 * ((unsigned)row * (unsigned)p.vs) + ch * 16;
      v_loff[i] = VBASE + row * (2 * D) + (((ch >> 2) ^ vx<D>(row)) << 6) + ((ch & 3) << 4);
    }
  }
  constexpr bool K_PARTIAL = (KBYTES / 16) % NT != 0;  // D=64: 256 chunks for 512 threads
  constexpr bool V_PARTIAL = (VBYTES / 16) % NT != 0;  // fp8 D=64
  const bool k_active = !K_PARTIAL || (t + NT * (KCH - 1)) < KBYTES / 16;
  const bool v_active = !V_PARTIAL || (t + NT * (VCH - 1)) < VBYTES / 16;
  // Fragment read addresses.  The swizzles depend only on the low row bits, so the block / k-step / high-half
  // parts are compile-time byte offsets folded into the ds_read immediates; per lane only KS (K) and DB or 4
  // (V) base registers are needed.
  unsigned kf_base[KS];  // + kb2 * 32 * D
#pragma unroll
  for (int s = 0; s < KS; ++s) kf_base[s] = r * D + (((2 * s + hh) ^ kx<D>(r)) << 4);
  constexpr int NVB = FP8 ? 4 : DB;
  unsigned vf_base[NVB];  // f16: [db] + ks*16*2D + hi*8*2D ;  fp8: [ks] + db*32*64
#pragma unroll
  for (int i = 0; i < NVB; ++i) {
    if constexpr (FP8) {
      vf_base[i] = VBASE + r * 64 + (((2 * i + hh) ^ ((r >> 2) & 7)) << 3);
    } else {
      const int vrow = 4 * hh + ((lane & 15) >> 2);
      const int vcol = 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
      vf_base[i] = VBASE + vrow * (2 * D) + ((i ^ vx<D>(vrow)) << 6) + vcol;
    }
  }

  // ---- staging ---------------------------------------------------------------------------------------------
  u32x4 kreg[KCH], vreg[VCH];
  auto load_tile = [&](int j) __attribute__((always_inline)) {  // rows / tiles past the end are outside the descriptor and read as zeros
    const int64_t ko = (int64_t)j * k_tile_stride, vo = (int64_t)j * v_tile_stride;
    const __amdgpu_buffer_rsrc_t k_rs = make_rsrc(kbase + ko, (unsigned)max((int64_t)0, k_bytes - ko));
    const __amdgpu_buffer_rsrc_t v_rs = make_rsrc(vbase + vo, (unsigned)max((int64_t)0, v_bytes - vo));
#pragma unroll
    for (int i = 0; i < KCH; ++i)
      if (i < KCH - 1 || k_active) kreg[i] = buf_load16(k_rs, k_goff[i], 0);
#pragma unroll
    for (int i = 0; i < VCH; ++i)
      if (i < VCH - 1 || v_active) vreg[i] = buf_load16(v_rs, v_goff[i], 0);
  };
  auto store_tile = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < KCH; ++i)
      if (i < KCH - 1 || k_active) *reinterpret_cast<u32x4*>(smem + k_loff[i] + slot * KBYTES) = kreg[i];
#pragma unroll
    for (int i = 0; i < VCH; ++i) {
      u32x4 val = vreg[i];
      if constexpr (VT == LBFA_BF16) {  // bf16 -> fp16 on the way in (src/core.py:307-308 `v.to(float16)`)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = __uint_as_float(val[e] << 16), hi = __uint_as_float(val[e] & 0xffff0000u);
          const f16x2 pk = f16x2{(_Float16)lo, (_Float16)hi};
          val[e] = __builtin_bit_cast(unsigned, pk);
        }
      }
      if (i < VCH - 1 || v_active) *reinterpret_cast<u32x4*>(smem + v_loff[i] + slot * VBYTES) = val;
    }
  };

  // ---- running state -----------------------------------------------------------------------------------------
  f32x16 acc_o[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc_o[db][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  float x[2][16];  // scores of the tile in flight (floats kMagic + s), then P in place
  typedef typename std::conditional<FP8, long, f16x8>::type pfrag_t;
  pfrag_t pf[4];   // packed P^T fragments of the last exponentiated tile (consumed by the next MFMA segment)

  i32x16 cmagic;
#pragma unroll
  for (int i = 0; i < 16; ++i) cmagic[i] = kMagicBits;

  // exact bias folding on a common power-of-two grid (derivation: attn_fwd.hip)
  float ks_max = 0.f;
  for (int i = lane; i < p.nK; i += 64) ks_max = fmaxf(ks_max, ksc[i]);
  ks_max = fmaxf(wave_max(ks_max), 1e-30f);
  const float sc_max = qsc * ks_max;
  const int gexp = (int)((__float_as_uint(1.25f * kMagic * sc_max) >> 23) & 0xff) - 127 + 1 - 21;
  const float G = __builtin_ldexpf(1.0f, gexp), invG = __builtin_ldexpf(1.0f, -gexp);
  const float g = __builtin_ldexpf(1.0f, gexp - 22), invg = __builtin_ldexpf(1.0f, 22 - gexp);
  auto grid_up = [&](float m) __attribute__((always_inline)) { return __builtin_ceilf(m * invG) * G; };
  constexpr float kPLimit = 32768.0f;

  // ---- segments ------------------------------------------------------------------------------------------------
  // The MFMA segment runs with ONE wave per SIMD issuing matrix work (its partner is in its VALU segment),
  // so LDS latency is not hidden by other waves: every fragment of a half-segment is requested up front
  // (distinct registers), then the MFMAs drain them in order.
  auto qk = [&](int kofs) __attribute__((always_inline)) {  // S^T = K Q^T for tile at K-ring byte offset kofs (scores as floats kMagic + s)
    i32x4 kf[2][KS];
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
      for (int s = 0; s < KS; ++s)
        kf[kb2][s] = *reinterpret_cast<const i32x4*>(smem + (kf_base[s] + kofs) + kb2 * 32 * D);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) {
      i32x16 sacc;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (s == 0) sacc = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf[kb2][s], qf[s], cmagic, 0, 0, 0);
        else sacc = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf[kb2][s], qf[s], sacc, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) x[kb2][i] = __int_as_float(sacc[i]);
    }
  };
  auto pv = [&](int vofs) __attribute__((always_inline)) {  // O^T += V^T P^T with the packed P of the previous VALU segment
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      typedef typename std::conditional<FP8, long, f16x8>::type vfrag_t;
      vfrag_t vf[2][DB];
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const int ks = 2 * half + k2;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          if constexpr (FP8) {
            vf[k2][db] = *reinterpret_cast<const long*>(smem + (vf_base[ks] + vofs) + db * 2048);
          } else {
            const f16x4 lo = lds_read_tr16(smem + (vf_base[db] + vofs) + ks * 32 * D);
            const f16x4 hi = lds_read_tr16(smem + (vf_base[db] + vofs) + ks * 32 * D + 16 * D);
            vf[k2][db] = f16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const int ks = 2 * half + k2;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          if constexpr (FP8) acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(vf[k2][db], pf[ks], acc_o[db], 0, 0, 0);
          else acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[k2][db], pf[ks], acc_o[db], 0, 0, 0);
        }
      }
    }
  };
  auto update_reference = [&](float sc, float c0, float thr) __attribute__((always_inline)) {
    float tmax = -INFINITY;
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
      for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, x[kb2][i]);
    tmax = half_swap_max(tmax);
    const float xmax = __builtin_fmaf(tmax, sc, c0);  // row max of the dequantised scores; -inf if all masked
    const float m_cand = fmaxf(m_run, FP8 ? xmax : grid_up(xmax));
    if (__any(m_cand > m_run + thr)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_cand);  // m_run = -inf -> 0
      m_run = m_cand;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc_o[db][i] *= alpha;
    }
  };
  auto exponentiate = [&](float sc, float c0) __attribute__((always_inline)) -> float {
    float c1 = c0 - m_run;
    if constexpr (FP8) c1 += kFp8Offset;
    // all exponentials first (independent, throughput-bound), then the row sum with four independent partial
    // sums: a single wave per SIMD has nobody to hide a v_exp -> v_add dependency stall behind
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
      for (int i = 0; i < 16; ++i) x[kb2][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[kb2][i], sc, c1));
    __builtin_amdgcn_sched_barrier(0);
    float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
      for (int i = 0; i < 16; ++i) ps[i & 3] += x[kb2][i];
    return (ps[0] + ps[1]) + (ps[2] + ps[3]);
  };
  auto pack_p = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kb2 = ks >> 1, rb = (ks & 1) * 8;
      if constexpr (FP8) {
        unsigned w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(x[kb2][rb + 0], x[kb2][rb + 1], w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(x[kb2][rb + 2], x[kb2][rb + 3], w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(x[kb2][rb + 4], x[kb2][rb + 5], w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(x[kb2][rb + 6], x[kb2][rb + 7], w1, true);
        pf[ks] = (long)(((unsigned long)w1 << 32) | (unsigned long)w0);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[ks][e] = (_Float16)x[kb2][rb + e];
      }
    }
  };
  // VALU segment for tile j (scores already in x): softmax + pack.  mode: 0 = plain, 1 = masked
  auto valu_segment = [&](int j, auto masked_tag) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const float sc = __builtin_rintf(qsc * ksc[j] * invg) * g, c0 = -kMagic * sc;
    if constexpr (MASKED) {
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = j * 64 + 32 * kb2 + (i & 3) + 8 * (i >> 2) + 4 * hh;
          bool dead = key >= p.Sk;
          if constexpr (CAUSAL) dead = dead || (key > qrow);
          if (dead) x[kb2][i] = -INFINITY;  // fma(-inf, sc, c1) = -inf -> p = 0
        }
    }
    if constexpr (MASKED || FP8) {
      update_reference(sc, c0, 0.0f);
      l_run += exponentiate(sc, c0);
    } else {
      // lazy reference: exponentiate against the current reference; only if a row sum blew up (first tile:
      // reference = -inf -> +inf) recompute the scores, take the row max, move the reference and redo.
      float psum = exponentiate(sc, c0);
      if (__any(!(psum <= kPLimit))) {
        qk((j % NBUF) * KBYTES);
        update_reference(sc, c0, 0.0f);
        psum = exponentiate(sc, c0);
      }
      l_run += psum;
    }
    pack_p();
  };

  // ---- prologue: tiles 0 and 1 into the ring, tile 2 staged -----------------------------------------------
  load_tile(0);
  store_tile(0);
  load_tile(1);
  store_tile(1);
  load_tile(2);

  // per-wave tile classification (wave-uniform): skip = every key above every row of this wave (causal);
  // masked = some key needs -inf (causal diagonal region, ragged last tile)
  auto tile_skip = [&](int j) __attribute__((always_inline)) { return CAUSAL && (j * 64 > row0 + 31); };
  auto tile_masked = [&](int j) __attribute__((always_inline)) { return (CAUSAL && (j * 64 + 63 > row0)) || (ragged && j == p.nK - 1); };

  // ---- step loops.  Both groups execute the same barrier sequence (two per tile period); group 0 runs
  // {MFMA segment, VALU segment}, group 1 {VALU segment, MFMA segment}, so the partner waves on a SIMD are
  // always in complementary segments.  Each group has its own straight-line loop body.
  // Ring upkeep happens at the start of the second half-step of period t (global step 2t+1): tile t+1 takes
  // the slot of tile t-2, and the fetch of tile t+2 is issued.
  auto ring_upkeep = [&](int tper) __attribute__((always_inline)) {
    if (tper >= 1) {
      store_tile((tper + 1) % NBUF);
      load_tile(tper + 2);
    }
  };
  auto mfma_segment = [&](int tq, int tp) __attribute__((always_inline)) {  // PV of tile tp (if exponentiated), then QK of tile tq
#if LBFA_PP_PRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#ifndef LBFA_PP_NO_MFMA
    if (tp >= 0 && tp < n_tiles && !tile_skip(tp)) pv((tp % NBUF) * VBYTES);
    if (tq < n_tiles && !tile_skip(tq)) qk((tq % NBUF) * KBYTES);
#endif
#if LBFA_PP_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  auto valu_seg = [&](int tv) __attribute__((always_inline)) {
#ifndef LBFA_PP_NO_VALU
    if (tv >= 0 && tv < n_tiles && !tile_skip(tv)) {
      if (tile_masked(tv)) valu_segment(tv, std::true_type{});
      else valu_segment(tv, std::false_type{});
    }
#endif
  };
  // n_main = leading tiles that are plain (no mask, no skip) for every wave of the workgroup: the steady-state
  // loops below are branch-free over them; the remaining periods run the generic (conditional) segments.
  int n_main = n_tiles;
  if constexpr (CAUSAL) n_main = min(n_tiles, 4 * qt2);
  else if (ragged) n_main = n_tiles - 1;
  auto mfma_plain = [&](int tq) __attribute__((always_inline)) {  // PV(tq-1) then QK(tq), both unconditional
#if LBFA_PP_PRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#ifndef LBFA_PP_NO_MFMA
    pv(((tq - 1) % NBUF) * VBYTES);
    qk((tq % NBUF) * KBYTES);
#endif
#if LBFA_PP_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  // One loop for both groups: group 1 enters it one barrier late (and group 0 leaves it one barrier late), so
  // group 1's MFMA segment coincides with group 0's VALU segment.  Global step of a barrier = its ordinal; the
  // ring upkeep belongs to the odd global steps, which is the VALU half-period for group 0 and the MFMA
  // half-period for group 1.
  if (grp == 1) __syncthreads();
  int tper = 0;
  // (the upkeep sits at the END of the odd half-step: the staged tile has had more than a full period to
  // arrive, and the store completes before the next barrier)
  if (n_main >= 2) {
    __syncthreads();
    mfma_segment(0, -1);
    if (grp == 1) ring_upkeep(0);
    __syncthreads();
    valu_seg(0);
    if (grp == 0) ring_upkeep(0);
#ifdef LBFA_PP_STAMP
    long long tb1 = 0, tm_ = 0, tb2 = 0, tv_ = 0;
#define STAMP() __builtin_amdgcn_s_memtime()
#endif
    for (tper = 1; tper < n_main; ++tper) {
#ifdef LBFA_PP_STAMP
      long long t0 = STAMP();
#endif
      __syncthreads();
#ifdef LBFA_PP_STAMP
      long long t1 = STAMP();
#endif
      mfma_plain(tper);
      if (grp == 1) ring_upkeep(tper);
#ifdef LBFA_PP_STAMP
      asm volatile("" :: "v"(x[0][0]), "v"(x[1][15]));
      long long t2 = STAMP();
#endif
      __syncthreads();
#ifdef LBFA_PP_STAMP
      long long t3 = STAMP();
#endif
#ifndef LBFA_PP_NO_VALU
      valu_segment(tper, std::false_type{});
#endif
      if (grp == 0) ring_upkeep(tper);
#ifdef LBFA_PP_STAMP
      asm volatile("" :: "v"(pf[3]));
      long long t4 = STAMP();
      tb1 += t1 - t0; tm_ += t2 - t1; tb2 += t3 - t2; tv_ += t4 - t3;
#endif
    }
#ifdef LBFA_PP_STAMP
    if (blockIdx.x == 100 && lane == 0 && (wave == 0 || wave == 4))
      printf("wave %d tiles %d: barrier1 %lld mfma %lld barrier2 %lld valu %lld (cycles per tile)\n", wave, n_main - 1,
             tb1 / (n_main - 1), tm_ / (n_main - 1), tb2 / (n_main - 1), tv_ / (n_main - 1));
#endif
  }
  for (; tper <= n_tiles; ++tper) {
    __syncthreads();
    mfma_segment(tper, tper - 1);
    if (grp == 1) ring_upkeep(tper);
    __syncthreads();
    valu_seg(tper);
    if (grp == 0) ring_upkeep(tper);
  }
  if (grp == 0) __syncthreads();

  // ---- epilogue: O = O^T / l (x v_scale), LSE ----------------------------------------------------------------------
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
      float ls = log2f(l_tot) + m_run;
      if constexpr (FP8) ls -= kFp8Offset;
      p.lse[((int64_t)b * p.Hq + h) * p.Sq + qrow] = ls;
    }
  }
}

hipError_t launch_attn_fwd_pp(const AttnParams& p, int D, int v_dtype, int o_dtype, int causal, hipStream_t stream) {
  const unsigned n = (unsigned)p.B * p.Hq * ((p.Sq + 255) / 256);
  dim3 grid(n), block(512);
#define LBFA_A(DD, VT, OT)                                                                           \
  do {                                                                                               \
    if (causal) hipLaunchKernelGGL((attn_fwd_pp_kernel<DD, VT, OT, true>), grid, block, 0, stream, p);  \
    else hipLaunchKernelGGL((attn_fwd_pp_kernel<DD, VT, OT, false>), grid, block, 0, stream, p);        \
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
