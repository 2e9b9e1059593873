// Software-pipelined variant of the fused low-bit attention forward (gfx950).
//
// Same decomposition and arithmetic as attn_fwd.hip (one workgroup = 4 waves = one 128-row Q block, a wave
// owns 32 query rows, S^T = K Q^T biased by 1.5*2^23, exact single-fma exponent, lazy softmax reference,
// O^T += V^T P^T).  What changes is WHEN things happen inside a wave.  In attn_fwd.hip a wave alternates between
// an MFMA-only phase (QK^T, then PV) and a VALU-only phase (the exponentials): the two pipes overlap only across
// waves.  Here every wave runs a three-stage software pipeline over the key tiles,
//
//      tile j:    VALU   softmax(j)            (exponentials of the scores computed one tile earlier)
//                 MFMA   QK^T(j+1)             (next tile's scores)
//                 MFMA   PV(j-1)               (previous tile's probabilities, packed one tile earlier)
//
// and the body of a tile is emitted as an explicit interleave: one MFMA, its operand fetch a few steps ahead, and a
// slice of the exponentials per step, with a scheduling barrier between steps so the compiler keeps the order.
// An MFMA blocks the issue port for 8 of its 32 cycles; the rest of the gap is filled with this wave's own VALU work.
//
// LDS: K and V rings of three 64-key tiles each; K runs one tile ahead of V.  Iteration j reads K(j+1) [QK^T],
// V(j-1) [PV] and, only in the rare reference-overflow redo, K(j); it writes K(j+2) and V(j+1) (fetched from HBM
// during iteration j-1) into the slots of K(j-1) / V(j-2).  One workgroup barrier per tile.
// The loop is unrolled by six (3 ring phases x 2 score-register phases) so that every LDS offset is an immediate
// and no register copies are needed; the remaining main tiles and the masked tiles (causal diagonal, ragged
// tail) run a plain sequential body.
#include "attn_common.h"

namespace lbfa {

#ifndef LBFA_SP_PD
#define LBFA_SP_PD 3  // operand prefetch distance, in MFMA steps
#endif

template <int D, int VT, int OT, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_fwd_sp_kernel(AttnParams p) {
  constexpr bool FP8 = (VT == LBFA_E4M3);
  constexpr int KS = D / 32;
  constexpr int DB = D / 32;
  constexpr int KBYTES = 64 * D;
  constexpr int VBYTES = FP8 ? 64 * D : 128 * D;
  constexpr int KCH = KBYTES / (256 * 16);
  constexpr int VCH = VBYTES / (256 * 16);
  constexpr int VBASE = 3 * KBYTES;  // LDS: [K ring x3][V ring x3]
  __shared__ __attribute__((aligned(16))) char smem[3 * (KBYTES + VBYTES)];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, hh = lane >> 5;

  const unsigned w_id = xcd_remap(blockIdx.x, gridDim.x);
  int qt = (int)(w_id % (unsigned)p.nQ);
  const int bh = (int)(w_id / (unsigned)p.nQ);
  if constexpr (CAUSAL) qt = p.nQ - 1 - qt;
  const int b = bh / p.Hq, h = bh % p.Hq, hk = h / p.group;
  const int row0 = qt * 128 + wave * 32;
  const int qrow = row0 + r;

  // ---- operand windows ------------------------------------------------------------------------------------
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
  const float qsc = p.q_scale[((int64_t)b * p.Hq + h) * p.nQ + qt];
  const float* ksc = p.k_scale + ((int64_t)b * p.Hkv + hk) * p.nK;

  int n_tiles = p.nK;
  if constexpr (CAUSAL) n_tiles = min(p.nK, 2 * (qt + 1));
  int n_main = n_tiles;  // leading tiles that need no mask for any wave of the workgroup
  if constexpr (CAUSAL) n_main = min(n_tiles, 2 * qt);
  else if ((p.Sk & 63) != 0) n_main = n_tiles - 1;

  // ---- loop-invariant per-thread offsets -----------------------------------------------------------------
  unsigned k_goff[KCH], k_loff[KCH], v_goff[VCH], v_loff[VCH];
#pragma unroll
  for (int i = 0; i < KCH; ++i) {
    const int c = t + 256 * i, row = c / (D / 16), ch = c % (D / 16);
    k_goff[i] = (unsigned)row * (unsigned)p.ks + ch * 16;
    k_loff[i] = row * D + ((ch ^ kx<D>(row)) << 4);
  }
#pragma unroll
  for (int i = 0; i < VCH; ++i) {
    if constexpr (FP8) {
      v_goff[i] = (t + 256 * i) * 16;
      v_loff[i] = VBASE + (t + 256 * i) * 16;
    } else {
      const int c = t + 256 * i, row = c / (D / 8), ch = c % (D / 8);
      v_goff[i] = 2 * ((unsigned)row * (unsigned)p.vs) + ch * 16;
      v_loff[i] = VBASE + row * (2 * D) + (((ch >> 2) ^ vx<D>(row)) << 6) + ((ch & 3) << 4);
    }
  }
  unsigned kf_base[KS];  // + slot*KBYTES + kb2*32*D
#pragma unroll
  for (int s = 0; s < KS; ++s) kf_base[s] = r * D + (((2 * s + hh) ^ kx<D>(r)) << 4);
  constexpr int NVB = FP8 ? 4 : DB;
  unsigned vf_base[NVB];  // f16: [db] + slot*VBYTES + ks*32*D + hi*16*D ;  fp8: [ks] + slot*VBYTES + db*2048
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

  // ---- staging: K and V tiles travel separately (K one tile ahead of V) -----------------------------------
  u32x4 kreg[KCH], vreg[VCH];
  auto load_k = [&](int j) __attribute__((always_inline)) {
    const int64_t ko = (int64_t)j * k_tile_stride;
    const __amdgpu_buffer_rsrc_t k_rs = make_rsrc(kbase + ko, (unsigned)max((int64_t)0, k_bytes - ko));
#pragma unroll
    for (int i = 0; i < KCH; ++i) kreg[i] = buf_load16(k_rs, k_goff[i], 0);
  };
  auto load_v = [&](int j) __attribute__((always_inline)) {
    const int64_t vo = (int64_t)j * v_tile_stride;
    const __amdgpu_buffer_rsrc_t v_rs = make_rsrc(vbase + vo, (unsigned)max((int64_t)0, v_bytes - vo));
#pragma unroll
    for (int i = 0; i < VCH; ++i) vreg[i] = buf_load16(v_rs, v_goff[i], 0);
  };
  auto store_k = [&](int kofs) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < KCH; ++i) *reinterpret_cast<u32x4*>(smem + k_loff[i] + kofs) = kreg[i];
  };
  auto store_v = [&](int vofs) __attribute__((always_inline)) {
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
      *reinterpret_cast<u32x4*>(smem + v_loff[i] + vofs) = val;
    }
  };

  // ---- running state -----------------------------------------------------------------------------------------
  f32x16 acc_o[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc_o[db][i] = 0.f;
  float m_run = -INFINITY;  // softmax reference (base-2 domain), identical in both halves of a row
  float l_run = 0.f;        // running sum over THIS lane's keys (halves are added in the epilogue)
  float xa[2][16], xb[2][16];  // score / probability registers of two consecutive tiles (roles alternate)
  typedef typename std::conditional<FP8, long, f16x8>::type pfrag_t;
  pfrag_t pf[4];               // packed P^T fragments of the previous tile, consumed by its PV one tile later
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    if constexpr (FP8) pf[ks] = 0;
    else pf[ks] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
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
  auto tile_scale = [&](int j) __attribute__((always_inline)) { return __builtin_rintf(qsc * ksc[j] * invg) * g; };

  // ---- plain building blocks (sequential paths: prologue, redo, leftover and masked tiles) -------------------
  auto qk_into = [&](float (&x)[2][16], int kofs) __attribute__((always_inline)) {
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) {
      i32x16 sacc;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const i32x4 kf = *reinterpret_cast<const i32x4*>(smem + (kf_base[s] + kofs) + kb2 * 32 * D);
        if (s == 0) sacc = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf, qf[s], cmagic, 0, 0, 0);
        else sacc = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf, qf[s], sacc, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) x[kb2][i] = __int_as_float(sacc[i]);
    }
  };
  auto pv_from = [&](int vofs) __attribute__((always_inline)) {  // O^T += V^T P^T with the packed P in pf
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        if constexpr (FP8) {
          const long vf = *reinterpret_cast<const long*>(smem + (vf_base[ks] + vofs) + db * 2048);
          acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(vf, pf[ks], acc_o[db], 0, 0, 0);
        } else {
          const f16x4 lo = lds_read_tr16(smem + (vf_base[db] + vofs) + ks * 32 * D);
          const f16x4 hi = lds_read_tr16(smem + (vf_base[db] + vofs) + ks * 32 * D + 16 * D);
          const f16x8 vf = f16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[ks], acc_o[db], 0, 0, 0);
        }
      }
  };
  auto mask_scores = [&](float (&x)[2][16], int j) __attribute__((always_inline)) {
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = j * 64 + 32 * kb2 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        bool dead = key >= p.Sk;
        if constexpr (CAUSAL) dead = dead || (key > qrow);
        if (dead) x[kb2][i] = -INFINITY;  // fma(-inf, sc, c1) = -inf -> p = 0
      }
  };
  // returns the factor applied to everything accumulated so far (1 if the reference did not move)
  auto update_reference = [&](const float (&x)[2][16], float sc, float c0) __attribute__((always_inline)) -> float {
    float tmax = -INFINITY;
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
      for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, x[kb2][i]);
    tmax = half_swap_max(tmax);
    const float xmax = __builtin_fmaf(tmax, sc, c0);  // row max of the dequantised scores; -inf if all masked
    const float m_cand = fmaxf(m_run, FP8 ? xmax : grid_up(xmax));
    float alpha = 1.0f;
    if (__any(m_cand > m_run)) {
      alpha = __builtin_amdgcn_exp2f(m_run - m_cand);  // m_run = -inf -> 0
      m_run = m_cand;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc_o[db][i] *= alpha;
    }
    return alpha;
  };
  auto exponentiate = [&](float (&x)[2][16], float sc, float c0) __attribute__((always_inline)) -> float {
    float c1 = c0 - m_run;
    if constexpr (FP8) c1 += kFp8Offset;
    float psum = 0.f;
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        x[kb2][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[kb2][i], sc, c1));
        psum += x[kb2][i];
      }
    return psum;
  };
  auto pack_p = [&](const float (&x)[2][16]) __attribute__((always_inline)) {
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

#ifdef LBFA_SP_STAMP
  long long tstamp[4] = {0, 0, 0, 0};
#endif
  // ---- the pipelined tile body ---------------------------------------------------------------------------------
  // J3 = j % 3 (ring phase), CUR/NXT = score register sets of tile j / j+1.
  constexpr int NQK = 2 * KS;          // MFMAs of QK^T(j+1)
  constexpr int NPV = 4 * DB;          // MFMAs of PV(j-1)
  constexpr int NM = NQK + NPV;        // MFMA steps per tile
  constexpr int PD = LBFA_SP_PD;
  typedef typename std::conditional<FP8, long, f16x8>::type vfrag_t;

  auto body = [&](auto j3_tag, float (&cur)[2][16], float (&nxt)[2][16], int j) __attribute__((always_inline)) {
    constexpr int J3 = decltype(j3_tag)::value;
    constexpr int K_NEXT = ((J3 + 1) % 3) * KBYTES;   // K(j+1)
    constexpr int K_CUR = J3 * KBYTES;                // K(j)   (redo only)
    constexpr int V_PREV = ((J3 + 2) % 3) * VBYTES;   // V(j-1)
    constexpr int K_WR = ((J3 + 2) % 3) * KBYTES;     // K(j+2) -> slot of K(j-1)
    constexpr int V_WR = ((J3 + 1) % 3) * VBYTES;     // V(j+1) -> slot of V(j-2)
#ifdef LBFA_SP_STAMP
    const long long st0 = __builtin_amdgcn_s_memtime();
#endif
    // staged tiles (fetched during the previous iteration) go into the ring; next fetches are issued
    store_k(K_WR);
    store_v(V_WR);
    load_k(j + 3);
    load_v(j + 2);

#ifdef LBFA_SP_STAMP
    const long long st1 = __builtin_amdgcn_s_memtime();
#endif
    const float sc = tile_scale(j), c0 = -kMagic * sc;
    float c1 = c0 - m_run;  // exact (grid argument); +inf while m_run = -inf
    if constexpr (FP8) c1 += kFp8Offset;  // (the reference already covers this tile: see the end of the body)

    // operands of the MFMA steps, requested PD steps ahead of their use
    i32x4 kfrag[NQK];
    vfrag_t vfrag[NPV];
    f16x4 vlo[NPV], vhi[NPV];
    auto request = [&](auto i_tag) __attribute__((always_inline)) {
      constexpr int i = decltype(i_tag)::value;
      if constexpr (i < NQK) {
        constexpr int kb2 = i / KS, s = i % KS;
        kfrag[i] = *reinterpret_cast<const i32x4*>(smem + (kf_base[s] + K_NEXT) + kb2 * 32 * D);
      } else if constexpr (i < NM) {
        constexpr int idx = i - NQK, ks = idx / DB, db = idx % DB;
        if constexpr (FP8) {
          vfrag[idx] = *reinterpret_cast<const long*>(smem + (vf_base[ks] + V_PREV) + db * 2048);
        } else {
          vlo[idx] = lds_read_tr16(smem + (vf_base[db] + V_PREV) + ks * 32 * D);
          vhi[idx] = lds_read_tr16(smem + (vf_base[db] + V_PREV) + ks * 32 * D + 16 * D);
        }
      }
    };
    static_for<0, PD>([&](auto i) __attribute__((always_inline)) { request(i); });

    i32x16 sacc[2];
    float psum = 0.f;
    static_for<0, NM>([&](auto i_tag) __attribute__((always_inline)) {
      constexpr int i = decltype(i_tag)::value;
      request(std::integral_constant<int, i + PD>{});
      // -- one MFMA
      if constexpr (i < NQK) {
        constexpr int kb2 = i / KS, s = i % KS;
        if constexpr (s == 0) sacc[kb2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(kfrag[i], qf[s], cmagic, 0, 0, 0);
        else sacc[kb2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(kfrag[i], qf[s], sacc[kb2], 0, 0, 0);
      } else {
        constexpr int idx = i - NQK, ks = idx / DB, db = idx % DB;
        if constexpr (FP8) {
          acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(vfrag[idx], pf[ks], acc_o[db], 0, 0, 0);
        } else {
          const f16x8 vf = f16x8{vlo[idx][0], vlo[idx][1], vlo[idx][2], vlo[idx][3], vhi[idx][0], vhi[idx][1], vhi[idx][2], vhi[idx][3]};
          acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[ks], acc_o[db], 0, 0, 0);
        }
      }
      // -- a slice of this tile's exponentials: elements [e0, e1) of the 32 this lane owns
      constexpr int e0 = (32 * i) / NM, e1 = (32 * (i + 1)) / NM;
#pragma unroll
      for (int e = e0; e < e1; ++e) {
        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(cur[e >> 4][e & 15], sc, c1));
        cur[e >> 4][e & 15] = pv;
        psum += pv;
      }
      __builtin_amdgcn_sched_barrier(0);
    });
#ifdef LBFA_SP_STAMP
    asm volatile("" :: "v"(psum), "v"(sacc[1][15]), "v"(acc_o[DB - 1][15]));
    const long long st2 = __builtin_amdgcn_s_memtime();
#endif
    // next tile's scores: the accumulator bits are the floats kMagic + s
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
      for (int i = 0; i < 16; ++i) nxt[kb2][i] = __int_as_float(sacc[kb2][i]);

    if constexpr (!FP8) {
      // lazy reference: only if a row sum blew up (first tile: reference = -inf -> +inf) recompute this tile's
      // scores (K(j) is still in the ring), take the row max, move the reference and exponentiate again
      if (__any(!(psum <= kPLimit))) {
        qk_into(cur, K_CUR);
        update_reference(cur, sc, c0);
        psum = exponentiate(cur, sc, c0);
      }
    }
    l_run += psum;
    if constexpr (FP8) {
      // fp8 P is pinned to P_max = 448, so the exact row max of the NEXT tile must be the reference before that
      // tile is exponentiated.  O so far (PV(j-1) included - its MFMAs were issued above) and l are rescaled
      // inside; this tile's P, not yet multiplied into O, takes the same factor before it is packed.
      const float scn = tile_scale(j + 1);
      const float alpha = update_reference(nxt, scn, -kMagic * scn);
      if (__any(alpha != 1.0f)) {
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
          for (int i = 0; i < 16; ++i) cur[kb2][i] *= alpha;
      }
    }
    pack_p(cur);
#ifdef LBFA_SP_STAMP
    asm volatile("" :: "v"(pf[3]));
    const long long st3 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
#ifdef LBFA_SP_STAMP
    const long long st4 = __builtin_amdgcn_s_memtime();
    tstamp[0] += st1 - st0; tstamp[1] += st2 - st1; tstamp[2] += st3 - st2; tstamp[3] += st4 - st3;
#endif
  };

  // ---- sequential tile (leftover main tiles, masked tiles): QK, softmax with the exact row max, PV ----------------
  auto plain_tile = [&](int j, bool masked) __attribute__((always_inline)) {
    const int s3 = j % 3;
    // ring upkeep identical to the pipelined body
    store_k(((j + 2) % 3) * KBYTES);
    store_v(((j + 1) % 3) * VBYTES);
    load_k(j + 3);
    load_v(j + 2);
    bool skip = false;
    if constexpr (CAUSAL) skip = j * 64 > row0 + 31;  // every key of the tile is above every row of this wave
    if (!skip) {
      const float sc = tile_scale(j), c0 = -kMagic * sc;
      qk_into(xa, s3 * KBYTES);
      if (masked) mask_scores(xa, j);
      update_reference(xa, sc, c0);
      l_run += exponentiate(xa, sc, c0);
      pack_p(xa);
      pv_from(s3 * VBYTES);
    }
    __syncthreads();
  };

  // ---- prologue -----------------------------------------------------------------------------------------------
  // ring: K(0), K(1), V(0) resident; K(2), V(1) staged in registers; V slot 2 zeroed (read by the empty PV(-1))
  load_k(0);
  load_v(0);
  store_k(0);
  store_v(0);
  load_k(1);
  store_k(KBYTES);
  for (int i = t; i < VBYTES / 16; i += 256) *reinterpret_cast<u32x4*>(smem + VBASE + 2 * VBYTES + i * 16) = u32x4{0, 0, 0, 0};
  load_k(2);
  load_v(1);
  __syncthreads();

  using T0 = std::integral_constant<int, 0>;
  using T1 = std::integral_constant<int, 1>;
  using T2 = std::integral_constant<int, 2>;
  int j = 0;
  const int n_fast = (n_main / 6) * 6;  // tiles run by the pipelined loop
  if (n_fast > 0) {
    qk_into(xa, 0);  // scores of tile 0
    if constexpr (FP8) {
      const float sc0 = tile_scale(0);
      update_reference(xa, sc0, -kMagic * sc0);
    }
    for (; j < n_fast; j += 6) {
      body(T0{}, xa, xb, j);
      body(T1{}, xb, xa, j + 1);
      body(T2{}, xa, xb, j + 2);
      body(T0{}, xb, xa, j + 3);
      body(T1{}, xa, xb, j + 4);
      body(T2{}, xb, xa, j + 5);
    }
#ifdef LBFA_SP_STAMP
    if (blockIdx.x == 100 && lane == 0 && (wave == 0 || wave == 3))
      printf("wave %d tiles %d: upkeep %lld steps %lld tail(pack/redo) %lld barrier %lld (cycles per tile)\n", wave, n_fast,
             tstamp[0] / n_fast, tstamp[1] / n_fast, tstamp[2] / n_fast, tstamp[3] / n_fast);
#endif
    // drain: PV of the last pipelined tile (its V is in slot (n_fast-1) % 3 = 2)
    pv_from(2 * VBYTES);
  }
  for (; j < n_tiles; ++j) plain_tile(j, j >= n_main);

  // ---- epilogue: O = O^T / l (x v_scale), LSE ---------------------------------------------------------------------
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

hipError_t launch_attn_fwd_sp(const AttnParams& p, int D, int v_dtype, int o_dtype, int causal, hipStream_t stream) {
  const unsigned n = (unsigned)p.B * p.Hq * p.nQ;
  dim3 grid(n), block(256);
#define LBFA_A(DD, VT, OT)                                                                           \
  do {                                                                                               \
    if (causal) hipLaunchKernelGGL((attn_fwd_sp_kernel<DD, VT, OT, true>), grid, block, 0, stream, p);  \
    else hipLaunchKernelGGL((attn_fwd_sp_kernel<DD, VT, OT, false>), grid, block, 0, stream, p);        \
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
