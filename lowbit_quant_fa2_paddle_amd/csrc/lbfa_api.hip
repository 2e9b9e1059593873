// extern "C" entry points of liblowbit_fa_hip.so (declared in include/lowbit_fa.h).
// Argument validation mirrors the reference's checks (src/core.py:269-290, csrc/utils.cuh:19-37,
// csrc/dispatch_utils.h:23-34); failures return a status code and leave a thread-local message.
#include <cstdarg>
#include <cstdio>

#include "lbfa_common.h"

namespace lbfa {
// defined in quant_kernels.hip / attn_fwd.hip
int mean_rows_per_split(int S);
size_t v_fp8_payload_bytes(int B, int H, int S, int D);
hipError_t launch_mean_seq(const void* x, int dtype, void* out, void* ws, int B, int H, int S, int D, int d_valid,
                           const int64_t* st, hipStream_t stream, bool finalize);
hipError_t launch_quant_per_block(const QuantParams& p, int dtype, int D, int blk, hipStream_t stream);
hipError_t launch_quant_v_fp8(const void* v, int dtype, uint8_t* out, float* v_scale, int B, int H, int S, int D,
                              int d_valid, const int64_t* st, hipStream_t stream);
hipError_t launch_attn_fwd(const AttnParams& p, int D, int v_dtype, int o_dtype, int causal, hipStream_t stream);
hipError_t launch_attn_fwd_f16(const AttnParams& p, int D, int dtype, int causal, hipStream_t stream);
hipError_t launch_attn_fwd_qq(const AttnParams& p, int D, int dtype, int v_dtype, int causal, hipStream_t stream);
hipError_t launch_cast_bf16_f16(const void* src, void* dst, int B, int H, int S, int d_valid, const int64_t* ss, const int64_t* ds,
                                hipStream_t stream);
hipError_t launch_absmax(const void* x, int dtype, float* out, int B, int H, int S, int D, const int64_t* st, hipStream_t stream);
}  // namespace lbfa

namespace {
thread_local char g_err[512] = "";
thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;  // one-shot, see lbfa_profile_next_attn

// launches the fused attention kernel, bracketed by the caller's events when a profile request is pending
hipError_t launch_attention(const lbfa::AttnParams& p, int D, int v_dtype, int o_dtype, int causal, hipStream_t stream,
                            bool quantise_q = false);

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return LBFA_OK;
  return fail(LBFA_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
}
bool dims_ok(int B, int H, int S, int D) { return B > 0 && H > 0 && S > 0 && (D == 64 || D == 128); }
bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
// q_scale [B,Hq,nQ], k_scale [B,Hkv,nK], no packed-batch tables
void dense_scale_layout(lbfa::AttnParams& p) {
  p.qsc_b = (int64_t)p.Hq * p.nQ; p.qsc_h = p.nQ; p.qsc_blk = 1;
  p.ksc_b = (int64_t)p.Hkv * p.nK; p.ksc_h = p.nK; p.ksc_blk = 1;
  p.cu_q = p.cu_k = p.cu_qscale = p.cu_kscale = nullptr;
  p.qk_scale = 0.f;
  p.q_sm_scale = 0.f; p.q_qmax = 127.f; p.q_dot_vec = nullptr;
}
// head dims the one-call operators take directly: the kernels work on 64 / 128 channels and treat the rest as the
// zero padding of src/core.py:277-287 (never read, never written)
bool head_dim_ok(int D) { return D >= 8 && D <= 128 && D % 8 == 0; }
int padded_head_dim(int D) { return D <= 64 ? 64 : 128; }
// The PV product of the int8 operators runs on fp16 MFMAs (the reference casts, src/core.py:307-308): a bf16 V is cast once, in a
// pre-pass, into the workspace (4 bytes per element of HBM traffic: 1..6 % of the operator from S = 4K up; converting the tiles on
// their way into LDS instead cost the attention kernel 9..10 % and 25..56 spilled registers).
bool v_cast_prepass(int dtype, int D_padded, int pv_fp8) { (void)D_padded; return dtype == LBFA_BF16 && !pv_fp8; }
}  // namespace

namespace {
hipError_t launch_attention(const lbfa::AttnParams& p, int D, int v_dtype, int o_dtype, int causal, hipStream_t stream,
                            bool quantise_q) {
  const hipEvent_t e0 = g_prof_start, e1 = g_prof_stop;
  g_prof_start = g_prof_stop = nullptr;
  if (e0) (void)hipEventRecord(e0, stream);
  const hipError_t err = quantise_q ? lbfa::launch_attn_fwd_qq(p, D, o_dtype, v_dtype, causal, stream)
                                    : lbfa::launch_attn_fwd(p, D, v_dtype, o_dtype, causal, stream);
  if (e1) (void)hipEventRecord(e1, stream);
  return err;
}
}  // namespace

extern "C" {

int lbfa_profile_next_attn(void* start_event, void* stop_event) {
  g_prof_start = (hipEvent_t)start_event;
  g_prof_stop = (hipEvent_t)stop_event;
  return LBFA_OK;
}

int lbfa_version(void) { return LBFA_VERSION; }

int lbfa_absmax(const void* x, int dtype, float* out, int B, int H, int S, int D, const int64_t strides_x[3], void* stream) {
  if (!x || !out || !strides_x) return fail(LBFA_EINVAL, "lbfa_absmax: null pointer");
  if (B <= 0 || H <= 0 || S <= 0 || D <= 0 || D % 8 != 0) return fail(LBFA_EINVAL, "lbfa_absmax: empty tensor or head_dim %d not a multiple of 8", D);
  if (dtype != LBFA_F16 && dtype != LBFA_BF16)
    return fail(LBFA_EINVAL, "Input tensors must be in dtype of float16 or bfloat16");
  if (!aligned16(x) || (strides_x[0] | strides_x[1] | strides_x[2]) % 8 != 0)
    return fail(LBFA_EINVAL, "lbfa_absmax: x must be 16-byte aligned with strides that are multiples of 8 elements");
  g_err[0] = 0;
  return check_hip(lbfa::launch_absmax(x, dtype, out, B, H, S, D, strides_x, (hipStream_t)stream), "lbfa_absmax launch");
}

const char* lbfa_last_error(void) { return g_err; }

size_t lbfa_mean_seq_workspace_bytes(int B, int H, int S, int D) {
  if (B <= 0 || H <= 0 || S <= 0 || D <= 0) return 0;
  const int rps = lbfa::mean_rows_per_split(S);
  const size_t nsplit = (size_t)(S + rps - 1) / rps;
  return (size_t)B * H * nsplit * D * sizeof(double);
}

namespace {
int mean_impl(const void* x, int dtype, void* mean_out, void* workspace, size_t workspace_bytes,
              int B, int H, int S, int D, int d_valid, const int64_t strides_x[3], void* stream, bool finalize = true) {
  if (!x || !mean_out || !workspace || !strides_x) return fail(LBFA_EINVAL, "lbfa_mean_seq: null pointer");
  if (!dims_ok(B, H, S, D)) return fail(LBFA_EINVAL, "Unsupported head_dim: %d (or empty tensor %dx%dx%d)", D, B, H, S);
  if (dtype != LBFA_F16 && dtype != LBFA_BF16)
    return fail(LBFA_EINVAL, "Input tensors must be in dtype of float16 or bfloat16");
  if (workspace_bytes < lbfa_mean_seq_workspace_bytes(B, H, S, D)) return fail(LBFA_EINVAL, "lbfa_mean_seq: workspace too small");
  if (!aligned16(x) || (strides_x[0] | strides_x[1] | strides_x[2]) % 8 != 0)
    return fail(LBFA_EINVAL, "lbfa_mean_seq: x must be 16-byte aligned with strides that are multiples of 8 elements");
  g_err[0] = 0;
  return check_hip(lbfa::launch_mean_seq(x, dtype, mean_out, workspace, B, H, S, D, d_valid, strides_x, (hipStream_t)stream, finalize),
                   "lbfa_mean_seq launch");
}
}  // namespace

int lbfa_mean_seq(const void* x, int dtype, void* mean_out, void* workspace, size_t workspace_bytes,
                  int B, int H, int S, int D, const int64_t strides_x[3], void* stream) {
  return mean_impl(x, dtype, mean_out, workspace, workspace_bytes, B, H, S, D, D, strides_x, stream);
}

namespace {
int quant_impl(const void* x, int dtype, const void* mean, int mean_group, int8_t* out, float* scale,
               float sm_scale, int qmax, int blk, int B, int H, int S, int D, int d_valid,
               const int64_t strides_x[3], const int64_t strides_out[3],
               const void* rowdot_vec, int rowdot_group, float* rowdot_out, void* stream,
               const double* mean_partial = nullptr, int mean_nsplit = 0) {
  if (!x || !out || !scale || !strides_x || !strides_out) return fail(LBFA_EINVAL, "lbfa_quant_per_block: null pointer");
  if (!dims_ok(B, H, S, D)) return fail(LBFA_EINVAL, "Unsupported head_dim: %d (or empty tensor %dx%dx%d)", D, B, H, S);
  if (dtype != LBFA_F16 && dtype != LBFA_BF16)
    return fail(LBFA_EINVAL, "Input tensors must be in dtype of float16 or bfloat16");
  if (qmax != 127 && qmax != 7) return fail(LBFA_EINVAL, "lbfa_quant_per_block: qmax must be 127 (int8) or 7 (int4 range), got %d", qmax);
  if (blk != 128 && blk != 64) return fail(LBFA_EINVAL, "lbfa_quant_per_block: blk must be 128 or 64, got %d", blk);
  if (mean && (mean_group <= 0 || H % mean_group != 0)) return fail(LBFA_EINVAL, "lbfa_quant_per_block: bad mean_group %d for H=%d", mean_group, H);
  if (rowdot_vec && (!rowdot_out || rowdot_group <= 0 || H % rowdot_group != 0))
    return fail(LBFA_EINVAL, "lbfa_quant_per_block: rowdot_vec needs rowdot_out and a rowdot_group dividing H");
  if (!aligned16(x) || (strides_x[0] | strides_x[1] | strides_x[2]) % 8 != 0)
    return fail(LBFA_EINVAL, "lbfa_quant_per_block: x must be 16-byte aligned with strides that are multiples of 8 elements");
  if ((reinterpret_cast<uintptr_t>(out) & 7u) || (strides_out[0] | strides_out[1] | strides_out[2]) % 8 != 0)
    return fail(LBFA_EINVAL, "lbfa_quant_per_block: out must be 8-byte aligned with strides that are multiples of 8");
  lbfa::QuantParams p;
  p.x = (const unsigned short*)x;
  p.mean = (const unsigned short*)mean;
  p.out = out;
  p.scale = scale;
  p.rowdot_vec = (const unsigned short*)rowdot_vec;
  p.rowdot_out = rowdot_out;
  p.xb = strides_x[0]; p.xh = strides_x[1]; p.xs = strides_x[2];
  p.ob = strides_out[0]; p.oh = strides_out[1]; p.os = strides_out[2];
  p.sm_scale = sm_scale;
  p.qmax = (float)qmax;
  p.B = B; p.H = H; p.S = S;
  p.nblk = (S + blk - 1) / blk;
  p.mean_group = mean ? mean_group : 1;
  p.rowdot_group = rowdot_vec ? rowdot_group : 1;
  p.scale_b = (int64_t)H * p.nblk; p.scale_h = p.nblk; p.scale_blk = 1;
  p.cu_seqlens = nullptr; p.cu_scale = nullptr; p.mean_b = 1;
  p.d_valid = d_valid;
  // fused last step of the mean: `mean` is then the OUTPUT buffer km is stored to
  p.mean_partial = mean ? mean_partial : nullptr;
  p.mean_out = mean_partial ? (unsigned short*)const_cast<void*>(mean) : nullptr;
  p.mean_nsplit = mean_nsplit;
  p.mean_S = S;
  g_err[0] = 0;
  return check_hip(lbfa::launch_quant_per_block(p, dtype, D, blk, (hipStream_t)stream), "lbfa_quant_per_block launch");
}
}  // namespace

int lbfa_quant_per_block(const void* x, int dtype, const void* mean, int mean_group, int8_t* out, float* scale,
                         float sm_scale, int qmax, int blk, int B, int H, int S, int D,
                         const int64_t strides_x[3], const int64_t strides_out[3],
                         const void* rowdot_vec, int rowdot_group, float* rowdot_out, void* stream) {
  return quant_impl(x, dtype, mean, mean_group, out, scale, sm_scale, qmax, blk, B, H, S, D, D, strides_x, strides_out,
                    rowdot_vec, rowdot_group, rowdot_out, stream);
}

namespace {
// packed batch, shared by lbfa_quant_per_block_varlen (reference scale layout) and lbfa_forward_varlen (padded layout)
int quant_varlen_core(const char* who, const void* x, int dtype, const void* mean, int mean_group, int8_t* out, float* scale,
                      const int32_t* cu_seqlens, const int32_t* cu_scale, float sm_scale, int qmax, int blk, int B,
                      int max_seqlen, int H, int D, int d_valid, const int64_t strides_x[2], const int64_t strides_out[2],
                      void* stream) {
  if (!x || !out || !scale || !cu_seqlens || !strides_x || !strides_out) return fail(LBFA_EINVAL, "%s: null pointer", who);
  if (!dims_ok(B, H, max_seqlen, D)) return fail(LBFA_EINVAL, "Unsupported head_dim: %d (or empty batch %dx%dx%d)", D, B, H, max_seqlen);
  if (dtype != LBFA_F16 && dtype != LBFA_BF16)
    return fail(LBFA_EINVAL, "Input tensors must be in dtype of float16 or bfloat16");
  if (qmax != 127 && qmax != 7) return fail(LBFA_EINVAL, "%s: qmax must be 127 (int8) or 7 (int4 range), got %d", who, qmax);
  if (blk != 128 && blk != 64) return fail(LBFA_EINVAL, "%s: blk must be 128 or 64, got %d", who, blk);
  if (mean && (mean_group <= 0 || H % mean_group != 0)) return fail(LBFA_EINVAL, "%s: bad mean_group %d for H=%d", who, mean_group, H);
  if (!aligned16(x) || (strides_x[0] | strides_x[1]) % 8 != 0)
    return fail(LBFA_EINVAL, "%s: x must be 16-byte aligned with strides that are multiples of 8 elements", who);
  if ((reinterpret_cast<uintptr_t>(out) & 7u) || (strides_out[0] | strides_out[1]) % 8 != 0)
    return fail(LBFA_EINVAL, "%s: out must be 8-byte aligned with strides that are multiples of 8", who);
  lbfa::QuantParams p;
  p.x = (const unsigned short*)x;
  p.mean = (const unsigned short*)mean;
  p.out = out;
  p.scale = scale;
  p.rowdot_vec = nullptr;
  p.rowdot_out = nullptr;
  p.xb = 0; p.xh = strides_x[0]; p.xs = strides_x[1];
  p.ob = 0; p.oh = strides_out[0]; p.os = strides_out[1];
  p.sm_scale = sm_scale;
  p.qmax = (float)qmax;
  p.B = B; p.H = H; p.S = max_seqlen;
  p.nblk = (max_seqlen + blk - 1) / blk;  // grid extent; blocks past a sequence's end exit at once
  p.mean_group = mean ? mean_group : 1;
  p.rowdot_group = 1;
  if (cu_scale) { p.scale_b = H; p.scale_h = 1; p.scale_blk = H; }              // [sum_blocks, H]
  else { p.scale_b = (int64_t)H * p.nblk; p.scale_h = p.nblk; p.scale_blk = 1; }  // [B, H, max_blocks]
  p.cu_seqlens = cu_seqlens; p.cu_scale = cu_scale; p.mean_b = 0;
  p.d_valid = d_valid;
  p.mean_partial = nullptr; p.mean_out = nullptr; p.mean_nsplit = 0; p.mean_S = 0;
  g_err[0] = 0;
  return check_hip(lbfa::launch_quant_per_block(p, dtype, D, blk, (hipStream_t)stream), who);
}
}  // namespace

int lbfa_quant_per_block_varlen(const void* x, int dtype, const void* mean, int mean_group, int8_t* out, float* scale,
                                const int32_t* cu_seqlens, const int32_t* cu_seqlens_scale, float sm_scale, int qmax,
                                int blk, int B, int max_seqlen, int H, int D, const int64_t strides_x[2],
                                const int64_t strides_out[2], void* stream) {
  if (!cu_seqlens_scale) return fail(LBFA_EINVAL, "lbfa_quant_per_block_varlen: null pointer");
  return quant_varlen_core("lbfa_quant_per_block_varlen", x, dtype, mean, mean_group, out, scale, cu_seqlens, cu_seqlens_scale,
                           sm_scale, qmax, blk, B, max_seqlen, H, D, D, strides_x, strides_out, stream);
}

size_t lbfa_v_fp8_bytes(int B, int H, int S, int D) {
  if (B <= 0 || H <= 0 || S <= 0 || D <= 0) return 0;
  // payload [B,H,ceil(S/64),D,64] + fp32 amax scratch [B,H,D]
  return lbfa::v_fp8_payload_bytes(B, H, S, D) + (size_t)B * H * D * sizeof(float);
}

namespace {
int quant_v_fp8_impl(const void* v, int dtype, uint8_t* v_fp8, float* v_scale, int B, int H, int S, int D, int d_valid,
                     const int64_t strides_v[3], void* stream) {
  if (!v || !v_fp8 || !v_scale || !strides_v) return fail(LBFA_EINVAL, "lbfa_quant_v_fp8: null pointer");
  if (!dims_ok(B, H, S, D)) return fail(LBFA_EINVAL, "Unsupported head_dim: %d (or empty tensor %dx%dx%d)", D, B, H, S);
  if (dtype != LBFA_F16 && dtype != LBFA_BF16)
    return fail(LBFA_EINVAL, "Input tensors must be in dtype of float16 or bfloat16");
  if (!aligned16(v) || !aligned16(v_fp8) || (strides_v[0] | strides_v[1] | strides_v[2]) % 8 != 0)
    return fail(LBFA_EINVAL, "lbfa_quant_v_fp8: v / v_fp8 must be 16-byte aligned, strides multiples of 8 elements");
  g_err[0] = 0;
  return check_hip(lbfa::launch_quant_v_fp8(v, dtype, v_fp8, v_scale, B, H, S, D, d_valid, strides_v, (hipStream_t)stream),
                   "lbfa_quant_v_fp8 launch");
}
}  // namespace

int lbfa_quant_v_fp8(const void* v, int dtype, uint8_t* v_fp8, float* v_scale, int B, int H, int S, int D,
                     const int64_t strides_v[3], void* stream) {
  return quant_v_fp8_impl(v, dtype, v_fp8, v_scale, B, H, S, D, D, strides_v, stream);
}

int lbfa_cast_bf16_to_f16(const void* src, void* dst, int B, int H, int S, int D, const int64_t strides_src[3],
                          const int64_t strides_dst[3], void* stream) {
  if (!src || !dst || !strides_src || !strides_dst) return fail(LBFA_EINVAL, "lbfa_cast_bf16_to_f16: null pointer");
  if (B <= 0 || H <= 0 || S <= 0 || D <= 0 || D % 8 != 0) return fail(LBFA_EINVAL, "lbfa_cast_bf16_to_f16: bad shape (D must be a multiple of 8)");
  if (!aligned16(src) || !aligned16(dst) || (strides_src[0] | strides_src[1] | strides_src[2] | strides_dst[0] | strides_dst[1] | strides_dst[2]) % 8 != 0)
    return fail(LBFA_EINVAL, "lbfa_cast_bf16_to_f16: tensors must be 16-byte aligned, strides multiples of 8 elements");
  g_err[0] = 0;
  return check_hip(lbfa::launch_cast_bf16_f16(src, dst, B, H, S, D, strides_src, strides_dst, (hipStream_t)stream), "lbfa_cast_bf16_to_f16 launch");
}

int lbfa_attn_fwd(const int8_t* q, const int8_t* k, const void* v, int v_dtype, void* o, int o_dtype, float* lse,
                  const float* q_scale, const float* k_scale, const float* v_scale,
                  int B, int Hq, int Hkv, int Sq, int Sk, int D,
                  const int64_t strides_q[3], const int64_t strides_k[3], const int64_t strides_v[3],
                  const int64_t strides_o[3], int is_causal, void* stream) {
  if (!q || !k || !v || !o || !q_scale || !k_scale || !strides_q || !strides_k || !strides_o)
    return fail(LBFA_EINVAL, "lbfa_attn_fwd: null pointer");
  if (B <= 0 || Hq <= 0 || Hkv <= 0 || Sq <= 0 || Sk <= 0) return fail(LBFA_EINVAL, "lbfa_attn_fwd: empty tensor");
  if (D != 64 && D != 128) return fail(LBFA_EINVAL, "Unsupported head_dim: %d", D);
  if (Hq % Hkv != 0) return fail(LBFA_EINVAL, "num_qo_heads (%d) must be divisible by num_kv_heads (%d)", Hq, Hkv);
  if (v_dtype == LBFA_BF16)
    return fail(LBFA_EINVAL, "lbfa_attn_fwd: v must be float16 (cast bfloat16 with lbfa_cast_bf16_to_f16, the `v.to(float16)` of the reference) or e4m3");
  if (v_dtype != LBFA_F16 && v_dtype != LBFA_E4M3) return fail(LBFA_EINVAL, "lbfa_attn_fwd: bad v_dtype %d", v_dtype);
  if (o_dtype != LBFA_F16 && o_dtype != LBFA_BF16) return fail(LBFA_EINVAL, "lbfa_attn_fwd: bad o_dtype %d", o_dtype);
  if (v_dtype == LBFA_E4M3 && !v_scale) return fail(LBFA_EINVAL, "lbfa_attn_fwd: v_scale is required for fp8 V");
  if (v_dtype != LBFA_E4M3 && !strides_v) return fail(LBFA_EINVAL, "lbfa_attn_fwd: strides_v is required for fp16 V");
  if (is_causal && Sq != Sk) return fail(LBFA_EINVAL, "qo_len and kv_len must be equal for causal attention");
  if (!aligned16(q) || !aligned16(k) || !aligned16(v) || (reinterpret_cast<uintptr_t>(o) & 7u))
    return fail(LBFA_EINVAL, "lbfa_attn_fwd: q/k/v must be 16-byte aligned and o 8-byte aligned");
  if ((strides_q[0] | strides_q[1] | strides_q[2] | strides_k[0] | strides_k[1] | strides_k[2]) % 16 != 0)
    return fail(LBFA_EINVAL, "lbfa_attn_fwd: q/k strides must be multiples of 16 elements");
  if (v_dtype != LBFA_E4M3 && (strides_v[0] | strides_v[1] | strides_v[2]) % 8 != 0)
    return fail(LBFA_EINVAL, "lbfa_attn_fwd: v strides must be multiples of 8 elements");
  if ((strides_o[0] | strides_o[1] | strides_o[2]) % 4 != 0) return fail(LBFA_EINVAL, "lbfa_attn_fwd: o strides must be multiples of 4 elements");
  // 32-bit buffer offsets: every per-(batch, head) operand window (+ one tile of look-ahead) must stay below 2 GiB
  {
    const int64_t lim = 0x7fffffffLL;
    const int64_t qwin = ((int64_t)Sq + LBFA_BLKQ) * strides_q[2] + D;
    const int64_t kwin = ((int64_t)Sk + 2 * LBFA_BLKK) * strides_k[2] + D;
    const int64_t vwin = v_dtype == LBFA_E4M3 ? ((int64_t)Sk + 2 * LBFA_BLKK) * D
                                              : 2 * (((int64_t)Sk + 2 * LBFA_BLKK) * strides_v[2] + D);
    if (qwin > lim || kwin > lim || vwin > lim)
      return fail(LBFA_EINVAL, "lbfa_attn_fwd: per-(batch,head) operand window exceeds 2 GiB (seq stride x length too large)");
  }
  lbfa::AttnParams p;
  p.q = q; p.k = k; p.v = v; p.o = o; p.lse = lse;
  p.q_scale = q_scale; p.k_scale = k_scale; p.v_scale = v_scale;
  p.qb = strides_q[0]; p.qh = strides_q[1]; p.qs = strides_q[2];
  p.kb = strides_k[0]; p.kh = strides_k[1]; p.ks = strides_k[2];
  if (v_dtype != LBFA_E4M3) { p.vb = strides_v[0]; p.vh = strides_v[1]; p.vs = strides_v[2]; }
  else { p.vb = p.vh = p.vs = 0; }
  p.ob = strides_o[0]; p.oh = strides_o[1]; p.os = strides_o[2];
  p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.Sq = Sq; p.Sk = Sk;
  p.nQ = (Sq + LBFA_BLKQ - 1) / LBFA_BLKQ;
  p.nK = (Sk + LBFA_BLKK - 1) / LBFA_BLKK;
  p.group = Hq / Hkv;
  p.lse_corr = nullptr;
  p.lse_scale = 1.0f;
  p.lse_corr_scale = 0.0f;
  dense_scale_layout(p);
  p.d_valid = D;
  if ((int64_t)B * Hq * p.nQ > 0x7fffffffLL) return fail(LBFA_EINVAL, "lbfa_attn_fwd: grid too large");
  g_err[0] = 0;
  return check_hip(launch_attention(p, D, v_dtype, o_dtype, is_causal ? 1 : 0, (hipStream_t)stream), "lbfa_attn_fwd launch");
}


// ---------------------------------------------------------------------------------------------------------------
// whole operator in one call
// ---------------------------------------------------------------------------------------------------------------
namespace {
size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
struct FwdLayout {
  size_t km, part, q8, k8, qs, ks, corr, v8, vs, v16, total;
};
FwdLayout fwd_layout(int B, int Hq, int Hkv, int Sq, int Sk, int D, int pv_fp8, int want_corr, int dtype) {
  FwdLayout L;
  size_t o = 0;
  auto take = [&](size_t n) { size_t at = o; o += align256(n); return at; };
  L.km = take((size_t)B * Hkv * D * 2);
  L.part = take(lbfa_mean_seq_workspace_bytes(B, Hkv, Sk, D));
  // Q is quantised inside the attention kernel (no codes, scales or lse-correction buffers)
  (void)Hq; (void)Sq; (void)want_corr;
  L.q8 = L.qs = L.corr = 0;
  L.k8 = take((size_t)B * Hkv * Sk * D);
  L.ks = take((size_t)B * Hkv * ((Sk + LBFA_BLKK - 1) / LBFA_BLKK) * 4);
  L.v8 = take(pv_fp8 ? lbfa_v_fp8_bytes(B, Hkv, Sk, D) : 0);
  L.vs = take(pv_fp8 ? (size_t)B * Hkv * D * 4 : 0);
  L.v16 = take(v_cast_prepass(dtype, D, pv_fp8) ? (size_t)B * Hkv * Sk * D * 2 : 0);  // fp16 copy of a bf16 V, [B,Hkv,Sk,D]
  L.total = o;
  return L;
}
}  // namespace

size_t lbfa_forward_workspace_bytes_dt(int B, int Hq, int Hkv, int Sq, int Sk, int D, int dtype, int pv_fp8, int smooth_k, int return_lse) {
  if (B <= 0 || Hq <= 0 || Hkv <= 0 || Sq <= 0 || Sk <= 0 || !head_dim_ok(D)) return 0;
  return fwd_layout(B, Hq, Hkv, Sq, Sk, padded_head_dim(D), pv_fp8, smooth_k && return_lse, dtype).total;
}
size_t lbfa_forward_workspace_bytes(int B, int Hq, int Hkv, int Sq, int Sk, int D, int pv_fp8, int smooth_k, int return_lse) {
  return lbfa_forward_workspace_bytes_dt(B, Hq, Hkv, Sq, Sk, D, LBFA_BF16, pv_fp8, smooth_k, return_lse);  // enough for either dtype
}

int lbfa_forward(const void* q, const void* k, const void* v, int dtype, void* o, float* lse, void* workspace,
                 size_t workspace_bytes, int B, int Hq, int Hkv, int Sq, int Sk, int D,
                 const int64_t strides_q[3], const int64_t strides_k[3], const int64_t strides_v[3],
                 const int64_t strides_o[3], double sm_scale, int q_qmax, int k_qmax, int pv_fp8, int is_causal,
                 int smooth_k, void* stream) {
  const int Dg = D;  // head dim of the caller's tensors
  if (!q || !k || !v || !o || !workspace || !strides_q || !strides_k || !strides_v || !strides_o)
    return fail(LBFA_EINVAL, "lbfa_forward: null pointer");
  if (B <= 0 || Hq <= 0 || Hkv <= 0 || Sq <= 0 || Sk <= 0) return fail(LBFA_EINVAL, "lbfa_forward: empty tensor");
  if (!head_dim_ok(Dg)) return fail(LBFA_EINVAL, "Unsupported head_dim: %d", Dg);
  D = padded_head_dim(Dg);  // what the kernels run on; channels >= Dg are never read or written
  if (Hq % Hkv != 0) return fail(LBFA_EINVAL, "num_qo_heads (%d) must be divisible by num_kv_heads (%d)", Hq, Hkv);
  const int want_lse = lse != nullptr, want_corr = want_lse && smooth_k;
  const FwdLayout L = fwd_layout(B, Hq, Hkv, Sq, Sk, D, pv_fp8, want_corr, dtype);
  if (workspace_bytes < L.total) return fail(LBFA_EINVAL, "lbfa_forward: workspace too small (%zu < %zu)", workspace_bytes, L.total);
  if (!aligned16(workspace)) return fail(LBFA_EINVAL, "lbfa_forward: workspace must be 16-byte aligned");
  // Every argument check comes BEFORE the first launch: a call that returns LBFA_EINVAL has enqueued nothing (no wasted
  // work, no stray nodes in a stream capture the caller believes failed).
  if (dtype != LBFA_F16 && dtype != LBFA_BF16) return fail(LBFA_EINVAL, "Input tensors must be in dtype of float16 or bfloat16");
  if (q_qmax != 127 && q_qmax != 7) return fail(LBFA_EINVAL, "lbfa_forward: q_qmax must be 127 (int8) or 7 (int4 range), got %d", q_qmax);
  if (k_qmax != 127 && k_qmax != 7) return fail(LBFA_EINVAL, "lbfa_forward: k_qmax must be 127 (int8) or 7 (int4 range), got %d", k_qmax);
  if (!aligned16(q) || (strides_q[0] | strides_q[1] | strides_q[2]) % 8 != 0)
    return fail(LBFA_EINVAL, "lbfa_forward: q must be 16-byte aligned with strides that are multiples of 8 elements");
  if (!aligned16(k) || (strides_k[0] | strides_k[1] | strides_k[2]) % 8 != 0)
    return fail(LBFA_EINVAL, "lbfa_forward: k must be 16-byte aligned with strides that are multiples of 8 elements");
  if (!aligned16(v) || (strides_v[0] | strides_v[1] | strides_v[2]) % 8 != 0)
    return fail(LBFA_EINVAL, "lbfa_forward: v must be 16-byte aligned with strides that are multiples of 8 elements");
  if ((reinterpret_cast<uintptr_t>(o) & 7u) || (strides_o[0] | strides_o[1] | strides_o[2]) % 4 != 0)
    return fail(LBFA_EINVAL, "lbfa_forward: o must be 8-byte aligned with strides that are multiples of 4 elements");
  if (is_causal && Sq != Sk) return fail(LBFA_EINVAL, "qo_len and kv_len must be equal for causal attention");
  {
    const int64_t lim = 0x7fffffffLL;
    const int64_t vwin = pv_fp8 ? ((int64_t)Sk + 2 * LBFA_BLKK) * D : 2 * (((int64_t)Sk + 2 * LBFA_BLKK) * strides_v[2] + D);
    if (2 * (((int64_t)Sq + LBFA_BLKQ) * strides_q[2] + D) > lim || ((int64_t)Sk + 2 * LBFA_BLKK) * D + D > lim || vwin > lim)
      return fail(LBFA_EINVAL, "lbfa_forward: per-(batch,head) operand window exceeds 2 GiB");
  }
  if ((int64_t)B * Hq * ((Sq + LBFA_BLKQ - 1) / LBFA_BLKQ) > 0x7fffffffLL) return fail(LBFA_EINVAL, "lbfa_forward: grid too large");
  char* ws = (char*)workspace;
  void* km = smooth_k ? (void*)(ws + L.km) : nullptr;
  int8_t* k8 = (int8_t*)(ws + L.k8);
  float* ks = (float*)(ws + L.ks);
  const int64_t sk8[3] = {(int64_t)Hkv * Sk * D, (int64_t)Sk * D, D};
  int st;
  // smooth-K mean: fp64 partial sums per 256 / 1024 rows, and - while there are few of them - no separate finalize launch:
  // the K quantiser's workgroups add the partials of their (batch, head) themselves (same order, same roundings) and
  // store km for the LSE correction.  (mean_finalize alone is 6.6 us of pure launch latency at C2.)
  const int nsplit = (Sk + lbfa::mean_rows_per_split(Sk) - 1) / lbfa::mean_rows_per_split(Sk);
  const bool fuse_finalize = smooth_k && nsplit <= 64;
  if (smooth_k) {
    st = mean_impl(k, dtype, km, ws + L.part, lbfa_mean_seq_workspace_bytes(B, Hkv, Sk, D), B, Hkv, Sk, D, Dg, strides_k, stream,
                   !fuse_finalize);
    if (st) return st;
  }
  // Q is quantised by the attention kernel itself (each workgroup its own 128-row block: same codes and scales as
  // lbfa_quant_per_block, sm_scale * log2(e) folded in, src/triton/quant_per_block.py:226), as is lse_correction = q . km
  st = quant_impl(k, dtype, km, 1, k8, ks, 1.0f, k_qmax, LBFA_BLKK, B, Hkv, Sk, D, Dg, strides_k, sk8, nullptr, 1, nullptr, stream,
                  fuse_finalize ? (const double*)(ws + L.part) : nullptr, nsplit);
  if (st) return st;
  const void* v_in = v;
  int v_dtype = dtype;
  const float* v_scale = nullptr;
  if (pv_fp8) {
    st = quant_v_fp8_impl(v, dtype, (uint8_t*)(ws + L.v8), (float*)(ws + L.vs), B, Hkv, Sk, D, Dg, strides_v, stream);
    if (st) return st;
    v_in = ws + L.v8;
    v_dtype = LBFA_E4M3;
    v_scale = (const float*)(ws + L.vs);
  }
  int64_t sv[3] = {strides_v[0], strides_v[1], strides_v[2]};
  if (v_cast_prepass(dtype, D, pv_fp8)) {
    const int64_t sv16[3] = {(int64_t)Hkv * Sk * D, (int64_t)Sk * D, D};
    st = check_hip(lbfa::launch_cast_bf16_f16(v, ws + L.v16, B, Hkv, Sk, Dg, strides_v, sv16, (hipStream_t)stream), "lbfa_forward (V cast) launch");
    if (st) return st;
    v_in = ws + L.v16;
    v_dtype = LBFA_F16;
    sv[0] = sv16[0]; sv[1] = sv16[1]; sv[2] = sv16[2];
  }
  // attention, with the LSE fix-up of src/core.py:344-350 fused into the epilogue
  lbfa::AttnParams p;
  p.q = (const int8_t*)q; p.k = k8; p.v = v_in; p.o = o; p.lse = lse;
  p.q_scale = nullptr; p.k_scale = ks; p.v_scale = v_scale;
  p.qb = strides_q[0]; p.qh = strides_q[1]; p.qs = strides_q[2];
  p.kb = sk8[0]; p.kh = sk8[1]; p.ks = sk8[2];
  if (v_dtype != LBFA_E4M3) { p.vb = sv[0]; p.vh = sv[1]; p.vs = sv[2]; }
  else { p.vb = p.vh = p.vs = 0; }
  p.ob = strides_o[0]; p.oh = strides_o[1]; p.os = strides_o[2];
  p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.Sq = Sq; p.Sk = Sk;
  p.nQ = (Sq + LBFA_BLKQ - 1) / LBFA_BLKQ;
  p.nK = (Sk + LBFA_BLKK - 1) / LBFA_BLKK;
  p.group = Hq / Hkv;
  p.lse_corr = nullptr;
  p.lse_scale = 1.0f / 1.44269504f;   // natural-log LSE (src/core.py:347)
  p.lse_corr_scale = (float)sm_scale;
  dense_scale_layout(p);
  p.d_valid = Dg;
  p.q_sm_scale = (float)(sm_scale * 1.44269504);  // ONE rounding of the double product, as the reference's kernel argument
  p.q_qmax = (float)q_qmax;
  p.q_dot_vec = want_corr ? (const unsigned short*)km : nullptr;
  g_err[0] = 0;
  return check_hip(launch_attention(p, D, v_dtype, dtype, is_causal ? 1 : 0, (hipStream_t)stream, true), "lbfa_forward launch");
}


// ---------------------------------------------------------------------------------------------------------------
// un-quantised attention: the FP16 branch of the precision router (src/core.py:1066-1096 -> default_attn, :46-69)
// ---------------------------------------------------------------------------------------------------------------
int lbfa_sdpa_fwd(const void* q, const void* k, const void* v, int dtype, void* o, float* lse,
                  int B, int Hq, int Hkv, int Sq, int Sk, int D,
                  const int64_t strides_q[3], const int64_t strides_k[3], const int64_t strides_v[3],
                  const int64_t strides_o[3], double sm_scale, int is_causal, void* stream) {
  if (!q || !k || !v || !o || !strides_q || !strides_k || !strides_v || !strides_o) return fail(LBFA_EINVAL, "lbfa_sdpa_fwd: null pointer");
  if (B <= 0 || Hq <= 0 || Hkv <= 0 || Sq <= 0 || Sk <= 0) return fail(LBFA_EINVAL, "lbfa_sdpa_fwd: empty tensor");
  if (!head_dim_ok(D)) return fail(LBFA_EINVAL, "Unsupported head_dim: %d", D);
  if (Hq % Hkv != 0) return fail(LBFA_EINVAL, "num_qo_heads (%d) must be divisible by num_kv_heads (%d)", Hq, Hkv);
  if (dtype != LBFA_F16 && dtype != LBFA_BF16) return fail(LBFA_EINVAL, "Input tensors must be in dtype of float16 or bfloat16");
  if (is_causal && Sq != Sk) return fail(LBFA_EINVAL, "qo_len and kv_len must be equal for causal attention");
  if (!(sm_scale > 0.0)) return fail(LBFA_EINVAL, "lbfa_sdpa_fwd: sm_scale must be positive (the row maximum is tracked on the unscaled scores)");
  if (!aligned16(q) || !aligned16(k) || !aligned16(v) || (reinterpret_cast<uintptr_t>(o) & 7u))
    return fail(LBFA_EINVAL, "lbfa_sdpa_fwd: q/k/v must be 16-byte aligned and o 8-byte aligned");
  if ((strides_q[0] | strides_q[1] | strides_q[2] | strides_k[0] | strides_k[1] | strides_k[2] | strides_v[0] | strides_v[1] | strides_v[2]) % 8 != 0)
    return fail(LBFA_EINVAL, "lbfa_sdpa_fwd: q/k/v strides must be multiples of 8 elements");
  if ((strides_o[0] | strides_o[1] | strides_o[2]) % 4 != 0) return fail(LBFA_EINVAL, "lbfa_sdpa_fwd: o strides must be multiples of 4 elements");
  {
    const int64_t lim = 0x7fffffffLL;
    if (2 * (((int64_t)Sq + LBFA_BLKQ) * strides_q[2] + D) > lim || 2 * (((int64_t)Sk + 2 * LBFA_BLKK) * strides_k[2] + D) > lim ||
        2 * (((int64_t)Sk + 2 * LBFA_BLKK) * strides_v[2] + D) > lim)
      return fail(LBFA_EINVAL, "lbfa_sdpa_fwd: per-(batch,head) operand window exceeds 2 GiB");
  }
  lbfa::AttnParams p;
  p.q = (const int8_t*)q; p.k = (const int8_t*)k; p.v = v; p.o = o; p.lse = lse;
  p.q_scale = nullptr; p.k_scale = nullptr; p.v_scale = nullptr;
  p.qb = strides_q[0]; p.qh = strides_q[1]; p.qs = strides_q[2];
  p.kb = strides_k[0]; p.kh = strides_k[1]; p.ks = strides_k[2];
  p.vb = strides_v[0]; p.vh = strides_v[1]; p.vs = strides_v[2];
  p.ob = strides_o[0]; p.oh = strides_o[1]; p.os = strides_o[2];
  p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.Sq = Sq; p.Sk = Sk;
  p.nQ = (Sq + LBFA_BLKQ - 1) / LBFA_BLKQ;
  p.nK = (Sk + LBFA_BLKK - 1) / LBFA_BLKK;
  p.group = Hq / Hkv;
  p.lse_corr = nullptr;
  p.lse_scale = 1.0f / 1.44269504f;  // natural-log LSE
  p.lse_corr_scale = 0.0f;
  dense_scale_layout(p);
  p.d_valid = D;
  p.qk_scale = (float)(sm_scale * 1.44269504);
  if ((int64_t)B * Hq * p.nQ > 0x7fffffffLL) return fail(LBFA_EINVAL, "lbfa_sdpa_fwd: grid too large");
  g_err[0] = 0;
  const hipEvent_t e0 = g_prof_start, e1 = g_prof_stop;
  g_prof_start = g_prof_stop = nullptr;
  if (e0) (void)hipEventRecord(e0, (hipStream_t)stream);
  const hipError_t err = lbfa::launch_attn_fwd_f16(p, padded_head_dim(D), dtype, is_causal ? 1 : 0, (hipStream_t)stream);
  if (e1) (void)hipEventRecord(e1, (hipStream_t)stream);
  return check_hip(err, "lbfa_sdpa_fwd launch");
}

// ---------------------------------------------------------------------------------------------------------------
// packed variable-length batches (reference: sageattn_varlen, src/core.py:356-491)
// ---------------------------------------------------------------------------------------------------------------
namespace {
int attn_varlen_core(const char* who, const int8_t* q, const int8_t* k, const void* v, int v_dtype, void* o, int o_dtype,
                     const float* q_scale, const float* k_scale, const int32_t* cu_q, const int32_t* cu_k,
                     const int32_t* cu_qscale, const int32_t* cu_kscale, int B, int Hq, int Hkv, int max_q, int max_k, int D,
                     int d_valid, const int64_t sq[2], const int64_t sk[2], const int64_t sv[2], const int64_t so[2], int is_causal,
                     void* stream, bool quantise_q = false, float q_sm_scale = 0.f, int q_qmax = 127) {
  if (!q || !k || !v || !o || (!q_scale && !quantise_q) || !k_scale || !cu_q || !cu_k || !sq || !sk || !sv || !so)
    return fail(LBFA_EINVAL, "%s: null pointer", who);
  if ((cu_qscale == nullptr) != (cu_kscale == nullptr)) return fail(LBFA_EINVAL, "%s: give both scale offset tables or neither", who);
  if (B <= 0 || Hq <= 0 || Hkv <= 0 || max_q <= 0 || max_k <= 0) return fail(LBFA_EINVAL, "%s: empty batch", who);
  if (D != 64 && D != 128) return fail(LBFA_EINVAL, "Unsupported head_dim: %d", D);
  if (Hq % Hkv != 0) return fail(LBFA_EINVAL, "num_qo_heads (%d) must be divisible by num_kv_heads (%d)", Hq, Hkv);
  if (v_dtype != LBFA_F16) return fail(LBFA_EINVAL, "%s: v must be float16 (cast bfloat16 with lbfa_cast_bf16_to_f16)", who);
  if (o_dtype != LBFA_F16 && o_dtype != LBFA_BF16) return fail(LBFA_EINVAL, "%s: bad o_dtype %d", who, o_dtype);
  if (!aligned16(q) || !aligned16(k) || !aligned16(v) || (reinterpret_cast<uintptr_t>(o) & 7u))
    return fail(LBFA_EINVAL, "%s: q/k/v must be 16-byte aligned and o 8-byte aligned", who);
  if (((quantise_q ? 0 : (sq[0] | sq[1])) | sk[0] | sk[1]) % 16 != 0) return fail(LBFA_EINVAL, "%s: q/k strides must be multiples of 16 elements", who);
  if (quantise_q && (sq[0] | sq[1]) % 8 != 0) return fail(LBFA_EINVAL, "%s: q strides must be multiples of 8 elements", who);
  if (quantise_q && q_qmax != 127 && q_qmax != 7) return fail(LBFA_EINVAL, "%s: q_qmax must be 127 (int8) or 7 (int4 range), got %d", who, q_qmax);
  if ((sv[0] | sv[1]) % 8 != 0) return fail(LBFA_EINVAL, "%s: v strides must be multiples of 8 elements", who);
  if ((so[0] | so[1]) % 4 != 0) return fail(LBFA_EINVAL, "%s: o strides must be multiples of 4 elements", who);
  {
    const int64_t lim = 0x7fffffffLL;
    if ((quantise_q ? 2 : 1) * (((int64_t)max_q + LBFA_BLKQ) * sq[1] + D) > lim || ((int64_t)max_k + 2 * LBFA_BLKK) * sk[1] + D > lim ||
        2 * (((int64_t)max_k + 2 * LBFA_BLKK) * sv[1] + D) > lim)
      return fail(LBFA_EINVAL, "%s: per-sequence operand window exceeds 2 GiB (token stride x max_seqlen too large)", who);
  }
  lbfa::AttnParams p;
  p.q = q; p.k = k; p.v = v; p.o = o; p.lse = nullptr;
  p.q_scale = q_scale; p.k_scale = k_scale; p.v_scale = nullptr;
  p.qb = 0; p.qh = sq[0]; p.qs = sq[1];
  p.kb = 0; p.kh = sk[0]; p.ks = sk[1];
  p.vb = 0; p.vh = sv[0]; p.vs = sv[1];
  p.ob = 0; p.oh = so[0]; p.os = so[1];
  p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.Sq = max_q; p.Sk = max_k;
  p.nQ = (max_q + LBFA_BLKQ - 1) / LBFA_BLKQ;
  p.nK = (max_k + LBFA_BLKK - 1) / LBFA_BLKK;
  p.group = Hq / Hkv;
  p.lse_corr = nullptr; p.lse_scale = 1.0f; p.lse_corr_scale = 0.0f;
  dense_scale_layout(p);  // padded [B,H,max_blocks] unless the reference's packed tables are given
  if (cu_qscale) {
    p.qsc_b = Hq; p.qsc_h = 1; p.qsc_blk = Hq;      // [sum_q_blocks, Hq]  (attn_qk_int8_block_varlen.py:134-138)
    p.ksc_b = Hkv; p.ksc_h = 1; p.ksc_blk = Hkv;    // [sum_k_blocks, Hkv]
  }
  p.cu_q = cu_q; p.cu_k = cu_k; p.cu_qscale = cu_qscale; p.cu_kscale = cu_kscale;
  p.d_valid = d_valid;
  p.q_sm_scale = q_sm_scale; p.q_qmax = (float)q_qmax;
  if ((int64_t)B * Hq * p.nQ > 0x7fffffffLL) return fail(LBFA_EINVAL, "%s: grid too large", who);
  g_err[0] = 0;
  return check_hip(launch_attention(p, D, v_dtype, o_dtype, is_causal ? 1 : 0, (hipStream_t)stream, quantise_q), who);
}

struct VarlenLayout {
  size_t km, part, q8, k8, qs, ks, v16, total;
};
VarlenLayout varlen_layout(int B, int Hq, int Hkv, int total_q, int total_k, int max_q, int max_k, int D, int dtype) {
  VarlenLayout L;
  size_t o = 0;
  auto take = [&](size_t n) { size_t at = o; o += align256(n); return at; };
  L.km = take((size_t)Hkv * D * 2);
  L.part = take(lbfa_mean_seq_workspace_bytes(1, Hkv, total_k, D));
  (void)total_q; (void)Hq; (void)max_q;  // Q is quantised inside the attention kernel
  L.q8 = L.qs = 0;
  L.k8 = take((size_t)total_k * Hkv * D);
  L.ks = take((size_t)B * Hkv * ((max_k + LBFA_BLKK - 1) / LBFA_BLKK) * 4);
  L.v16 = take(v_cast_prepass(dtype, D, 0) ? (size_t)total_k * Hkv * D * 2 : 0);  // fp16 copy of a bf16 V, [total_k,Hkv,D]
  L.total = o;
  return L;
}
}  // namespace

int lbfa_attn_fwd_varlen(const int8_t* q, const int8_t* k, const void* v, int v_dtype, void* o, int o_dtype,
                         const float* q_scale, const float* k_scale, const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                         const int32_t* cu_seqlens_q_scale, const int32_t* cu_seqlens_k_scale,
                         int B, int Hq, int Hkv, int max_seqlen_q, int max_seqlen_k, int D,
                         const int64_t strides_q[2], const int64_t strides_k[2], const int64_t strides_v[2],
                         const int64_t strides_o[2], int is_causal, void* stream) {
  if (!cu_seqlens_q_scale || !cu_seqlens_k_scale) return fail(LBFA_EINVAL, "lbfa_attn_fwd_varlen: null pointer");
  return attn_varlen_core("lbfa_attn_fwd_varlen", q, k, v, v_dtype, o, o_dtype, q_scale, k_scale, cu_seqlens_q, cu_seqlens_k,
                          cu_seqlens_q_scale, cu_seqlens_k_scale, B, Hq, Hkv, max_seqlen_q, max_seqlen_k, D, D, strides_q, strides_k,
                          strides_v, strides_o, is_causal, stream);
}

size_t lbfa_forward_varlen_workspace_bytes_dt(int B, int Hq, int Hkv, int total_q, int total_k, int max_seqlen_q,
                                              int max_seqlen_k, int D, int dtype) {
  if (B <= 0 || Hq <= 0 || Hkv <= 0 || total_q <= 0 || total_k <= 0 || max_seqlen_q <= 0 || max_seqlen_k <= 0 || !head_dim_ok(D)) return 0;
  return varlen_layout(B, Hq, Hkv, total_q, total_k, max_seqlen_q, max_seqlen_k, padded_head_dim(D), dtype).total;
}
size_t lbfa_forward_varlen_workspace_bytes(int B, int Hq, int Hkv, int total_q, int total_k, int max_seqlen_q,
                                           int max_seqlen_k, int D) {
  return lbfa_forward_varlen_workspace_bytes_dt(B, Hq, Hkv, total_q, total_k, max_seqlen_q, max_seqlen_k, D, LBFA_BF16);  // enough for either dtype
}

int lbfa_forward_varlen(const void* q, const void* k, const void* v, int dtype, void* o,
                        const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k, void* workspace, size_t workspace_bytes,
                        int B, int Hq, int Hkv, int total_q, int total_k, int max_seqlen_q, int max_seqlen_k, int D,
                        const int64_t strides_q[2], const int64_t strides_k[2], const int64_t strides_v[2],
                        const int64_t strides_o[2], double sm_scale, int q_qmax, int k_qmax, int is_causal, int smooth_k,
                        void* stream) {
  const int Dg = D;  // head dim of the caller's tensors
  if (!q || !k || !v || !o || !workspace || !cu_seqlens_q || !cu_seqlens_k || !strides_q || !strides_k || !strides_v || !strides_o)
    return fail(LBFA_EINVAL, "lbfa_forward_varlen: null pointer");
  if (B <= 0 || Hq <= 0 || Hkv <= 0 || total_q <= 0 || total_k <= 0 || max_seqlen_q <= 0 || max_seqlen_k <= 0)
    return fail(LBFA_EINVAL, "lbfa_forward_varlen: empty batch");
  if (!head_dim_ok(Dg)) return fail(LBFA_EINVAL, "Unsupported head_dim: %d", Dg);
  D = padded_head_dim(Dg);
  if (Hq % Hkv != 0) return fail(LBFA_EINVAL, "num_qo_heads (%d) must be divisible by num_kv_heads (%d)", Hq, Hkv);
  const VarlenLayout L = varlen_layout(B, Hq, Hkv, total_q, total_k, max_seqlen_q, max_seqlen_k, D, dtype);
  if (workspace_bytes < L.total) return fail(LBFA_EINVAL, "lbfa_forward_varlen: workspace too small (%zu < %zu)", workspace_bytes, L.total);
  if (!aligned16(workspace)) return fail(LBFA_EINVAL, "lbfa_forward_varlen: workspace must be 16-byte aligned");
  // every argument check before the first launch (see lbfa_forward)
  if (dtype != LBFA_F16 && dtype != LBFA_BF16) return fail(LBFA_EINVAL, "Input tensors must be in dtype of float16 or bfloat16");
  if (q_qmax != 127 && q_qmax != 7) return fail(LBFA_EINVAL, "lbfa_forward_varlen: q_qmax must be 127 (int8) or 7 (int4 range), got %d", q_qmax);
  if (k_qmax != 127 && k_qmax != 7) return fail(LBFA_EINVAL, "lbfa_forward_varlen: k_qmax must be 127 (int8) or 7 (int4 range), got %d", k_qmax);
  if (!aligned16(q) || !aligned16(k) || !aligned16(v) || (reinterpret_cast<uintptr_t>(o) & 7u))
    return fail(LBFA_EINVAL, "lbfa_forward_varlen: q/k/v must be 16-byte aligned and o 8-byte aligned");
  if ((strides_q[0] | strides_q[1] | strides_k[0] | strides_k[1] | strides_v[0] | strides_v[1]) % 8 != 0)
    return fail(LBFA_EINVAL, "lbfa_forward_varlen: q / k / v strides must be multiples of 8 elements");
  if ((strides_o[0] | strides_o[1]) % 4 != 0) return fail(LBFA_EINVAL, "lbfa_forward_varlen: o strides must be multiples of 4 elements");
  {
    const int64_t lim = 0x7fffffffLL;
    if (2 * (((int64_t)max_seqlen_q + LBFA_BLKQ) * strides_q[1] + D) > lim || ((int64_t)max_seqlen_k + 2 * LBFA_BLKK) * Hkv * D + D > lim ||
        2 * (((int64_t)max_seqlen_k + 2 * LBFA_BLKK) * strides_v[1] + D) > lim)
      return fail(LBFA_EINVAL, "lbfa_forward_varlen: per-sequence operand window exceeds 2 GiB (token stride x max_seqlen too large)");
  }
  if ((int64_t)B * Hq * ((max_seqlen_q + LBFA_BLKQ - 1) / LBFA_BLKQ) > 0x7fffffffLL) return fail(LBFA_EINVAL, "lbfa_forward_varlen: grid too large");
  char* ws = (char*)workspace;
  void* km = smooth_k ? (void*)(ws + L.km) : nullptr;
  int8_t* k8 = (int8_t*)(ws + L.k8);
  float* ks = (float*)(ws + L.ks);
  const int64_t sk8[2] = {D, (int64_t)Hkv * D};
  int st;
  if (smooth_k) {  // km = k.mean(dim=0): over ALL tokens of the packed batch (src/core.py:453)
    const int64_t sk3[3] = {0, strides_k[0], strides_k[1]};
    st = mean_impl(k, dtype, km, ws + L.part, lbfa_mean_seq_workspace_bytes(1, Hkv, total_k, D), 1, Hkv, total_k, D, Dg, sk3, stream);
    if (st) return st;
  }
  // Q is quantised by the attention kernel itself (blocks restart at every sequence start, as the quantiser's do)
  st = quant_varlen_core("lbfa_forward_varlen (K)", k, dtype, km, 1, k8, ks, cu_seqlens_k, nullptr, 1.0f, k_qmax, LBFA_BLKK, B,
                         max_seqlen_k, Hkv, D, Dg, strides_k, sk8, stream);
  if (st) return st;
  const void* v_in = v;
  int v_dtype = dtype;
  int64_t sv[2] = {strides_v[0], strides_v[1]};
  if (v_cast_prepass(dtype, D, 0)) {
    const int64_t src3[3] = {0, strides_v[0], strides_v[1]}, dst3[3] = {0, D, (int64_t)Hkv * D};
    st = check_hip(lbfa::launch_cast_bf16_f16(v, ws + L.v16, 1, Hkv, total_k, Dg, src3, dst3, (hipStream_t)stream), "lbfa_forward_varlen (V cast) launch");
    if (st) return st;
    v_in = ws + L.v16;
    v_dtype = LBFA_F16;
    sv[0] = D; sv[1] = (int64_t)Hkv * D;
  }
  return attn_varlen_core("lbfa_forward_varlen", (const int8_t*)q, k8, v_in, v_dtype, o, dtype, nullptr, ks, cu_seqlens_q, cu_seqlens_k,
                          nullptr, nullptr, B, Hq, Hkv, max_seqlen_q, max_seqlen_k, D, Dg, strides_q, sk8, sv, strides_o,
                          is_causal, stream, true, (float)(sm_scale * 1.44269504), q_qmax);
}

}  // extern "C"
