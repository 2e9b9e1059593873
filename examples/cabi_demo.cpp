// Pure C/C++ host of the C ABI (include/lowbit_fa.h): no torch, no Python - what a Paddle custom op or any FFI would do.
// Allocates with hipMalloc, runs the whole operator with ONE lbfa_forward call and checks it against a naive fp32 SDPA
// computed on the host (the repo's manual_scaled_dot_product_attention, src/core.py:46-69).
//   hipcc --offload-arch=gfx950 -O2 examples/cabi_demo.cpp -Iinclude -Llowbit_quant_fa2_paddle_amd -llowbit_fa_hip \
//         -Wl,-rpath,$PWD/lowbit_quant_fa2_paddle_amd -o /tmp/cabi_demo && /tmp/cabi_demo
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lowbit_fa.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } \
  } while (0)

int main() {
  const int B = 1, Hq = 4, Hkv = 2, S = 333, D = 80;  // GQA, ragged S, head dim padded inside the kernels
  const int causal = 1;
  const size_t nq = (size_t)B * Hq * S * D, nk = (size_t)B * Hkv * S * D;
  std::vector<__half> q(nq), k(nk), v(nk);
  std::vector<float> qf(nq), kf(nk), vf(nk);
  unsigned s = 1u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (size_t i = 0; i < nq; ++i) { q[i] = __float2half(rnd()); qf[i] = __half2float(q[i]); }
  for (size_t i = 0; i < nk; ++i) { k[i] = __float2half(rnd() + 0.3f); kf[i] = __half2float(k[i]); }
  for (size_t i = 0; i < nk; ++i) { v[i] = __float2half(rnd()); vf[i] = __half2float(v[i]); }

  void *dq, *dk, *dv, *dout, *dws;
  float* dlse;
  CK(hipMalloc(&dq, nq * 2)); CK(hipMalloc(&dk, nk * 2)); CK(hipMalloc(&dv, nk * 2)); CK(hipMalloc(&dout, nq * 2));
  CK(hipMalloc(&dlse, (size_t)B * Hq * S * 4));
  CK(hipMemcpy(dq, q.data(), nq * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dk, k.data(), nk * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dv, v.data(), nk * 2, hipMemcpyHostToDevice));
  const size_t ws_bytes = lbfa_forward_workspace_bytes(B, Hq, Hkv, S, S, D, /*pv_fp8=*/0, /*smooth_k=*/1, /*return_lse=*/1);
  CK(hipMalloc(&dws, ws_bytes));
  hipStream_t stream;
  CK(hipStreamCreate(&stream));
  const int64_t sq[3] = {(int64_t)Hq * S * D, (int64_t)S * D, D};   // HND: {batch, head, seq} element strides
  const int64_t sk[3] = {(int64_t)Hkv * S * D, (int64_t)S * D, D};
  const double sm_scale = 1.0 / std::sqrt((double)D);  // a double, as the reference's Python float
  int st = lbfa_forward(dq, dk, dv, LBFA_F16, dout, dlse, dws, ws_bytes, B, Hq, Hkv, S, S, D, sq, sk, sk, sq, sm_scale, 127, 127,
                        /*pv_fp8=*/0, causal, /*smooth_k=*/1, stream);
  if (st != LBFA_OK) { fprintf(stderr, "lbfa_forward failed (%d): %s\n", st, lbfa_last_error()); return 1; }
  CK(hipStreamSynchronize(stream));
  std::vector<__half> o(nq);
  std::vector<float> lse((size_t)B * Hq * S);
  CK(hipMemcpy(o.data(), dout, nq * 2, hipMemcpyDeviceToHost));
  CK(hipMemcpy(lse.data(), dlse, lse.size() * 4, hipMemcpyDeviceToHost));

  // naive fp32 SDPA on the host
  double max_err = 0, max_lse_err = 0, mse = 0;
  std::vector<float> p(S);
  for (int h = 0; h < Hq; ++h) {
    const int hk = h / (Hq / Hkv);
    for (int i = 0; i < S; ++i) {
      float mx = -INFINITY;
      const int lim = causal ? i + 1 : S;
      for (int j = 0; j < lim; ++j) {
        float acc = 0;
        for (int d = 0; d < D; ++d) acc += qf[((size_t)h * S + i) * D + d] * kf[((size_t)hk * S + j) * D + d];
        p[j] = acc * (float)sm_scale;
        mx = std::fmax(mx, p[j]);
      }
      double l = 0;
      for (int j = 0; j < lim; ++j) { p[j] = std::exp(p[j] - mx); l += p[j]; }
      max_lse_err = std::fmax(max_lse_err, std::fabs((mx + std::log(l)) - lse[(size_t)h * S + i]));
      for (int d = 0; d < D; ++d) {
        double acc = 0;
        for (int j = 0; j < lim; ++j) acc += p[j] * vf[((size_t)hk * S + j) * D + d];
        const double ref = acc / l, got = __half2float(o[((size_t)h * S + i) * D + d]);
        max_err = std::fmax(max_err, std::fabs(ref - got));
        mse += (ref - got) * (ref - got);
      }
    }
  }
  mse /= (double)nq;
  printf("lbfa version %d: max|dO| = %.3e, MSE = %.3e, max|dLSE| = %.3e (int8 QK^T vs exact fp32 attention)\n", lbfa_version(), max_err, mse,
         max_lse_err);
  const bool ok = mse <= 1e-5 && max_lse_err <= 5e-2;  // the int8 path's accuracy on U(-1,1) inputs
  printf(ok ? "OK\n" : "FAILED\n");
  return ok ? 0 : 1;
}
