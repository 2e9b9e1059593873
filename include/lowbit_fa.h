/*
 * lowbit_fa.h - C ABI of liblowbit_fa_hip.so: low-bit FlashAttention-2 forward for MI355X (gfx950).
 *
 * This is the drop-in boundary for the reference's native layer.  The reference binds its native
 * code through two pybind11 modules that take torch tensors:
 *     _qattn  (12 functions, csrc/qattn/pybind.cpp:21-38)   - fused attention
 *     _fused  ( 8 functions, csrc/fused/pybind.cpp:21-33)   - quantisation / pre-processing
 * and, on the path that actually runs, through Triton launches made by Python host wrappers
 * (src/triton/quant_per_block.py:181-248, src/triton/attn_qk_int8_per_block.py:169-238,
 * src/triton/attn_qk_int8_per_block_causal.py:337-437).  The entry points below carry the same
 * information across a plain C boundary: raw device pointers, sizes, element strides, a stream.
 *
 * Conventions (same as the reference's native layer):
 *   - the CALLER allocates every output (csrc/qattn: `o`, src/quant.py:70-85: int8 buffers, scales);
 *   - layouts (HND / NHD) are expressed ONLY through strides, exactly as the Triton host wrappers do
 *     (attn_qk_int8_per_block.py:183-196); strides are in ELEMENTS: {batch, head, sequence};
 *     the last (head_dim) stride must be 1 (src/core.py:288-290);
 *   - every call only enqueues work on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream, which is where the reference launches, qk_int_sv_f8_cuda.cu:845); no host sync;
 *   - no process-global mutable state: safe to call concurrently on different devices / streams.  The only state the
 *     library keeps is per calling THREAD: the last error message and the one-shot event pair of
 *     lbfa_profile_next_attn (armed and consumed by the same thread);
 *   - the one-call entry points (lbfa_forward, lbfa_forward_varlen) validate every argument before their first launch: a
 *     call that returns LBFA_EINVAL has enqueued nothing (safe under stream capture);
 *   - return 0 on success, non-zero LBFA_E* on failure; `lbfa_last_error()` returns a thread-local
 *     message (the reference raises through TORCH_CHECK, csrc/utils.cuh:19-37, and
 *     std::invalid_argument for unsupported head_dim/flags, csrc/dispatch_utils.h:23-34).
 *
 * Quantisation granularity is the reference Triton path's: one fp32 scale per 128 query rows
 * (BLKQ) and per 64 key rows (BLKK) (src/triton/quant_per_block.py:182).
 */
#ifndef LOWBIT_FA_H_
#define LOWBIT_FA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LBFA_VERSION 200 /* 0.2.0 (see INTEGRATION.md "ABI versions"): + lbfa_absmax; the changes of 0.1.1 (lbfa_attn_fwd[_varlen] refuse a
                            bf16 V - cast it with lbfa_cast_bf16_to_f16 -, *_workspace_bytes without a dtype include the fp16 copy of V) are
                            incompatible with 0.1.0 callers and are versioned as such here */

/* element types */
#define LBFA_F16 0
#define LBFA_BF16 1
#define LBFA_E4M3 2 /* OCP e4m3fn (gfx950 native), only as v_dtype of lbfa_attn_fwd */

/* status codes */
#define LBFA_OK 0
#define LBFA_EINVAL 1   /* bad argument (unsupported head_dim, dtype, null pointer ...) */
#define LBFA_ELAUNCH 2  /* HIP launch / runtime error */

#define LBFA_BLKQ 128 /* query rows per scale   (src/triton/quant_per_block.py:182, attn BLOCK_M :179) */
#define LBFA_BLKK 64  /* key rows per scale     (src/triton/quant_per_block.py:182, attn BLOCK_N :180) */

int lbfa_version(void);

/* Thread-local description of the last failure on this thread ("" if none). */
const char* lbfa_last_error(void);

/*
 * Mean over the sequence of x[B,H,S,D] -> mean_out[B,H,D] (contiguous, same dtype as x).
 * Replaces `km = k.mean(dim=seq_dim, keepdim=True)` (src/core.py:292-293); fp64 accumulation in a
 * fixed order (deterministic): the correctly rounded mean (float64 mean -> fp32 -> storage dtype).
 * workspace: >= lbfa_mean_seq_workspace_bytes(B,H,S,D) bytes of device memory (fp64 partial sums), 16-byte aligned.
 */
size_t lbfa_mean_seq_workspace_bytes(int B, int H, int S, int D);
int lbfa_mean_seq(const void* x, int dtype, void* mean_out, void* workspace, size_t workspace_bytes,
                  int B, int H, int S, int D, const int64_t strides_x[3], void* stream);

/*
 * Per-block symmetric quantisation of x[B,H,S,D] (fp16/bf16) to int8 codes + one fp32 scale per
 * (b, h, block of `blk` rows).  Replaces the two launches of
 * `quant_per_block_int8_kernel` (src/triton/quant_per_block.py:132-178, launched :213-247) and, with
 * qmax = 7, `quant_per_block_int4_unpack_kernel` (:22-71) - one 4-bit-range value per int8 byte:
 *     x' = dtype(x - mean)            (only if mean != NULL; `k = k - km`, :186-187)
 *     xs = fp32(x') * sm_scale ; scale = max(max|xs|, 1e-7) / qmax
 *     code = trunc(xs / scale + 0.5*sign)      (round half away from zero, :174-176)
 * The 1e-7 floor is the CUDA quantiser's (csrc/fused/fused.cu:147); the Triton kernel has none and
 * produces NaN on an all-zero block.  Identical bits whenever max|xs| >= 1e-7.
 *   mean      : [B, H_mean, D] contiguous in x's dtype, or NULL.  H_mean = H / mean_group.
 *   out       : int8, strides_out (elements); scale: [B, H, ceil(S/blk)] contiguous fp32.
 *   rowdot_vec: optional [B, H/rowdot_group, D] (x's dtype).  When non-NULL, rowdot_out[B,H,S] (fp32,
 *               contiguous) receives dtype(sum_d x[b,h,s,d] * vec[b,h/rowdot_group,d]) of the
 *               UN-shifted, un-scaled x: the `lse_correction = q @ km^T` of src/core.py:294-304.
 *   qmax in {127, 7}; blk in {128, 64}; D in {64, 128}.
 */
int lbfa_quant_per_block(const void* x, int dtype, const void* mean, int mean_group, int8_t* out, float* scale,
                         float sm_scale, int qmax, int blk, int B, int H, int S, int D,
                         const int64_t strides_x[3], const int64_t strides_out[3],
                         const void* rowdot_vec, int rowdot_group, float* rowdot_out, void* stream);

/*
 * Per-channel FP8 quantisation of V[B,H,S,D] (fp16/bf16).  Replaces `per_channel_fp8`
 * (src/quant.py:210-291 -> TransposePadPermuteKernel + MeanScaleKernel, csrc/fused/fused.cu:263-428):
 *     v_scale[b,h,d] = max(max_s |v|, 1e-7) / 448 ; v_fp8 = e4m3fn_rn_sat(v * 448 / amax)
 * v_fp8 is written in the device layout lbfa_attn_fwd consumes for v_dtype = LBFA_E4M3:
 * [B, H, ceil(S/64), D, 64] bytes - per 64-key tile, channel-major, keys permuted for the MFMA
 * operand (the reference likewise transposes, pads to 64 and permutes for its mma fragment,
 * fused.cu:288-292; the permutation is a device detail, not API).
 *   v_fp8 bytes required: lbfa_v_fp8_bytes(B,H,S,D).  Padded keys are written as 0.
 */
size_t lbfa_v_fp8_bytes(int B, int H, int S, int D);
int lbfa_quant_v_fp8(const void* v, int dtype, uint8_t* v_fp8, float* v_scale, int B, int H, int S, int D,
                     const int64_t strides_v[3], void* stream);

/*
 * Fused attention forward: O = softmax(dequant(Q K^T)) V, FlashAttention-2 tiling, online softmax in
 * base 2 (sm_scale*log2e is already folded into q_scale by the quantiser).  Replaces
 * `_attn_fwd` (src/triton/attn_qk_int8_per_block.py:69-167), its causal twin
 * (src/triton/attn_qk_int8_per_block_causal.py:82-214), the int4 copies
 * (src/triton/quantization/attn_qk_int4_per_block{,_causal}.py) and, for v_dtype = LBFA_E4M3,
 * `qk_int8_sv_f8_accum_f32_fuse_v_scale_attn*` (csrc/qattn/qk_int_sv_f8_cuda.cu:46-692).
 *   q [B,Hq,Sq,D], k [B,Hkv,Sk,D] : int8 codes (any range within int8: 127 or 7)
 *   v : LBFA_F16 [B,Hkv,Sk,D] with strides_v - as the reference's kernel, which is only ever handed `v.to(float16)`
 *       (src/core.py:307-308); a bf16 V is cast with lbfa_cast_bf16_to_f16 first (the one-call entry points do that
 *       themselves) - or LBFA_E4M3 in the layout of lbfa_quant_v_fp8 (strides_v ignored) together with v_scale [B,Hkv,D].
 *   o : [B,Hq,Sq,D] fp16 or bf16 (o_dtype), strides_o.
 *   lse : NULL, or [B,Hq,Sq] contiguous fp32 receiving log2(l) + m (base-2 domain, exactly what the
 *         reference kernel stores, :164-167; the host converts it, src/core.py:344-350).
 *   q_scale [B,Hq,ceil(Sq/128)], k_scale [B,Hkv,ceil(Sk/64)] contiguous fp32.
 *   is_causal requires Sq == Sk (attn_qk_int8_per_block_causal.py:389).
 *   Keys >= Sk are masked with -inf (the CUDA path's behaviour, csrc/qattn/attn_utils.cuh:327-353; the
 *   Triton kernel lets them into the softmax as zeros, a defect that only shows when Sk % 64 != 0).
 *   D in {64, 128}; Hq % Hkv == 0.
 */
int lbfa_cast_bf16_to_f16(const void* src, void* dst, int B, int H, int S, int D, const int64_t strides_src[3],
                          const int64_t strides_dst[3], void* stream); /* src/core.py:307-308 `v.to(float16)`; D % 8 == 0 */
int lbfa_attn_fwd(const int8_t* q, const int8_t* k, const void* v, int v_dtype, void* o, int o_dtype, float* lse,
                  const float* q_scale, const float* k_scale, const float* v_scale,
                  int B, int Hq, int Hkv, int Sq, int Sk, int D,
                  const int64_t strides_q[3], const int64_t strides_k[3], const int64_t strides_v[3],
                  const int64_t strides_o[3], int is_causal, void* stream);

/*
 * max |x| of a [B,H,S,D] fp16 / bf16 view (strides {batch, head, seq} in elements, last dim contiguous, D % 8 == 0) into ONE
 * device float `out` (zeroed by the call, on `stream`).  Replaces the reduction of `compute_scale`, the per-tensor statistic of
 * the precision router (src/core.py:1039-1048: `paddle.compat.max(paddle.abs(tensor))`); dividing by 2^(bits-1) - 1 and the
 * thresholds of `select_quantization` (:1051-1063) stay on the host.  Order-independent (integer atomic max): deterministic.
 */
int lbfa_absmax(const void* x, int dtype, float* out, int B, int H, int S, int D, const int64_t strides_x[3], void* stream);

/*
 * Profiling aid: the NEXT fused-attention launch made on this thread (by lbfa_attn_fwd or lbfa_forward) is bracketed by
 * hipEventRecord(start_event) / hipEventRecord(stop_event) on its stream; one-shot.  Both are hipEvent_t passed
 * as void* (NULL clears).  Used by bench.py to time the dominant kernel inside the timed region.
 */
int lbfa_profile_next_attn(void* start_event, void* stop_event);

/*
 * The whole operator in ONE call: smooth-K mean, per-block quantisation of Q and K, (fp8: per-channel V
 * quantisation,) fused attention and the LSE fix-up - the body of `sageattn_qk_int8_pv_fp16_triton`
 * (src/core.py:292-350), of `sageattn_qk_int4_pv_fp16_triton` (q_qmax / k_qmax = 7) and of
 * `sageattn_qk_int8_pv_fp8_cuda` (pv_fp8 = 1) after their argument checks and head-dim padding.  Same kernels
 * and results as calling the entry points above one by one (the host front end uses this one to keep the
 * per-call host overhead at one FFI call and one workspace allocation).
 *   q, k, v : fp16/bf16 (dtype), strides in elements {batch, head, seq}; o same dtype, strides_o.
 *   lse     : NULL, or [B,Hq,Sq] fp32 receiving the NATURAL-log LSE incl. the smooth-K correction
 *             (lse2 / 1.44269504 + (q . km) * sm_scale, src/core.py:344-350).
 *   workspace: >= lbfa_forward_workspace_bytes_dt(..., dtype, ...) bytes, 16-byte aligned, caller-owned scratch (int8 codes,
 *             scales, km, partial sums, fp8 V, the fp16 copy of a bf16 V); contents are undefined
 *             afterwards.  lbfa_forward_workspace_bytes(...) (no dtype) returns the size that is enough for either dtype.
 *   sm_scale: softmax scale (1/sqrt(original head_dim) by default on the host side), a double as in the reference's
 *             Python: the Q quantiser multiplies by fp32(sm_scale * 1.44269504) formed in double
 *             (src/triton/quant_per_block.py:226) - with 4-bit-range codes a 1-ulp difference there flips codes;
 *             q_qmax/k_qmax in {127, 7}.
 *   D       : head dim of q, k, v, o - 64, 128, or ANY multiple of 8 up to 128: the kernels then work on 64 / 128
 *             channels and treat the missing ones as the zero padding of src/core.py:277-287 (never read, never
 *             written), so the host makes no padded copies; results are bit-identical to padding on the host.
 *             (lbfa_forward_varlen likewise; the modular entry points take D in {64, 128} only.)
 *   range   : scores s = q.k * sm_scale * log2(e) of any size with pv_fp8 (the kernel keeps its softmax reference as an exact
 *             product once it passes 2^16); with fp16 P (pv_fp8 = 0) up to |s| < 2^27 - q, k of a few thousand times N(0,1):
 *             beyond it the exponent's one fma `s sc - m` against the rounded reference can overflow fp16 P (rows of NaN), as
 *             in every FlashAttention-2 kernel, lbfa_sdpa_fwd included.
 */
size_t lbfa_forward_workspace_bytes(int B, int Hq, int Hkv, int Sq, int Sk, int D, int pv_fp8, int smooth_k, int return_lse);
size_t lbfa_forward_workspace_bytes_dt(int B, int Hq, int Hkv, int Sq, int Sk, int D, int dtype, int pv_fp8, int smooth_k,
                                       int return_lse);
int lbfa_forward(const void* q, const void* k, const void* v, int dtype, void* o, float* lse, void* workspace,
                 size_t workspace_bytes, int B, int Hq, int Hkv, int Sq, int Sk, int D,
                 const int64_t strides_q[3], const int64_t strides_k[3], const int64_t strides_v[3],
                 const int64_t strides_o[3], double sm_scale, int q_qmax, int k_qmax, int pv_fp8, int is_causal,
                 int smooth_k, void* stream);

/*
 * Un-quantised FlashAttention-2 forward, O = softmax(Q K^T * sm_scale) V, on the same tiling as lbfa_attn_fwd with
 * 16-bit MFMAs for both products in the dtype given: fp16 inputs -> fp16 Q K^T, fp16 P and V; bf16 inputs -> bf16 Q K^T, bf16 P
 * and V (no conversion anywhere, as a bf16 FlashAttention-2 does); fp32 softmax and accumulation.  This is the "FP16" branch of
 * the precision router `sageattn_multi_precision` (src/core.py:1066-1096), which the reference sends to `default_attn`
 * (:46-69 / the framework's SDPA).
 *   q [B,Hq,Sq,D], k / v [B,Hkv,Sk,D], o like q: one dtype (LBFA_F16 / LBFA_BF16), strides {batch, head, seq}.
 *   D: any multiple of 8 up to 128 (as lbfa_forward).  lse: NULL or [B,Hq,Sq] fp32 natural-log LSE.
 */
int lbfa_sdpa_fwd(const void* q, const void* k, const void* v, int dtype, void* o, float* lse,
                  int B, int Hq, int Hkv, int Sq, int Sk, int D,
                  const int64_t strides_q[3], const int64_t strides_k[3], const int64_t strides_v[3],
                  const int64_t strides_o[3], double sm_scale, int is_causal, void* stream);

/*
 * Packed variable-length batches (reference: `sageattn_varlen`, src/core.py:356-491).
 *   q [total_q, Hq, D], k / v [total_k, Hkv, D]; sequence b owns tokens [cu_seqlens[b], cu_seqlens[b+1]);
 *   cu_seqlens_* are int32 device arrays of B + 1 entries; strides are in elements {head, token}.
 *   max_seqlen_* must be UPPER BOUNDS of the sequence lengths (they size the grids and the padded scale rows, as in the
 *   reference); a sequence longer than the stated maximum is cut there and nothing is read outside the sized buffers.
 *   Quantisation blocks (128 query / 64 key rows) restart at the start of every sequence.
 *
 * lbfa_quant_per_block_varlen replaces one launch of the varlen `quant_per_block_int8_kernel`
 * (src/triton/quant_per_block_varlen.py:22-72,107-141): scale is [sum_b ceil(len_b/blk), H] with
 * cu_seqlens_scale[b] = number of blocks before sequence b (:92-106).  `mean` (NULL or [H/mean_group, D]) is ONE
 * vector per head shared by all sequences - the reference smooths with k.mean(dim=0) over the packed tokens
 * (src/core.py:452-454) - and is subtracted in the storage dtype as in the dense quantiser.
 *
 * lbfa_attn_fwd_varlen replaces `_attn_fwd` of src/triton/attn_qk_int8_block_varlen.py:94-197 and its causal twin
 * (attn_qk_int8_per_block_causal_varlen.py) with the same scale layout; causal assumes len_q == len_k per sequence.
 * No LSE (the reference returns none).  A sequence with len_k == 0 gets zeros.
 *
 * lbfa_forward_varlen = the whole operator in one call (mean over all tokens, both quantisers, attention) on a
 * caller-owned workspace of lbfa_forward_varlen_workspace_bytes(...) bytes; internally the scales use a padded
 * [B, H, max_blocks] layout, so no cu_seqlens_scale tables are needed.
 */
int lbfa_quant_per_block_varlen(const void* x, int dtype, const void* mean, int mean_group, int8_t* out, float* scale,
                                const int32_t* cu_seqlens, const int32_t* cu_seqlens_scale, float sm_scale, int qmax,
                                int blk, int B, int max_seqlen, int H, int D, const int64_t strides_x[2],
                                const int64_t strides_out[2], void* stream);
int lbfa_attn_fwd_varlen(const int8_t* q, const int8_t* k, const void* v, int v_dtype, void* o, int o_dtype,
                         const float* q_scale, const float* k_scale, const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                         const int32_t* cu_seqlens_q_scale, const int32_t* cu_seqlens_k_scale,
                         int B, int Hq, int Hkv, int max_seqlen_q, int max_seqlen_k, int D,
                         const int64_t strides_q[2], const int64_t strides_k[2], const int64_t strides_v[2],
                         const int64_t strides_o[2], int is_causal, void* stream);
size_t lbfa_forward_varlen_workspace_bytes(int B, int Hq, int Hkv, int total_q, int total_k, int max_seqlen_q,
                                           int max_seqlen_k, int D);
size_t lbfa_forward_varlen_workspace_bytes_dt(int B, int Hq, int Hkv, int total_q, int total_k, int max_seqlen_q,
                                              int max_seqlen_k, int D, int dtype);
int lbfa_forward_varlen(const void* q, const void* k, const void* v, int dtype, void* o,
                        const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k, void* workspace, size_t workspace_bytes,
                        int B, int Hq, int Hkv, int total_q, int total_k, int max_seqlen_q, int max_seqlen_k, int D,
                        const int64_t strides_q[2], const int64_t strides_k[2], const int64_t strides_v[2],
                        const int64_t strides_o[2], double sm_scale, int q_qmax, int k_qmax, int is_causal, int smooth_k,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LOWBIT_FA_H_ */
