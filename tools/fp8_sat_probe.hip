// Development probe: what v_cvt_pk_fp8_f32 returns beyond the e4m3fn range, with MODE.FP16_OVFL clear and set.
//   hipcc --offload-arch=gfx950 -O2 tools/fp8_sat_probe.hip -o /tmp/fp8_sat_probe && /tmp/fp8_sat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void probe(const float* x, unsigned* out, int n, int ovfl) {
  if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);  // hwreg(HW_REG_MODE, 23, 1) = FP16_OVFL
  const int i = threadIdx.x;
  if (i < n) out[i] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(x[i], -x[i], 0u, false);
}
int main() {
  const float h[] = {400.f, 447.9f, 448.f, 463.9f, 464.f, 464.1f, 480.f, 512.f, 1000.f, 65504.f, 3e38f, INFINITY, NAN};
  const int n = sizeof(h) / sizeof(h[0]);
  float* dx; unsigned* dout;
  hipMalloc(&dx, sizeof(h)); hipMalloc(&dout, n * 4);
  hipMemcpy(dx, h, sizeof(h), hipMemcpyHostToDevice);
  for (int ovfl = 0; ovfl < 2; ++ovfl) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dx, dout, n, ovfl);
    unsigned o[32]; hipMemcpy(o, dout, n * 4, hipMemcpyDeviceToHost);
    printf("FP16_OVFL=%d:", ovfl);
    for (int i = 0; i < n; ++i) printf("  %g -> +%02x / -%02x", h[i], o[i] & 0xff, (o[i] >> 8) & 0xff);
    printf("\n");
  }
  return 0;
}
