// Development probe: v_exp_f32 / fma / max on very large arguments (the un-quantised kernels see scores of any fp32 size).
//   hipcc --offload-arch=gfx950 -O2 tools/exp_probe.hip -o tools/exp_probe && tools/exp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void probe(const float* x, float* out, int n) {
  const int i = threadIdx.x;
  if (i < n) {
    out[i] = __builtin_amdgcn_exp2f(x[i]);
    out[n + i] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[i], 0.18f, -x[i] * 0.18f));
    out[2 * n + i] = __builtin_amdgcn_exp2f(-INFINITY - x[i]);
    out[3 * n + i] = __builtin_fmaf(x[i], 0.18f, INFINITY);
  }
}
int main() {
  const float h[] = {-1e3f, -1e6f, -2e9f, -2.2e9f, -7e10f, -1e20f, -3e38f, -INFINITY, 1e3f, 2e9f, 7e10f, 3e38f, INFINITY, 0.f, -126.f, -150.f};
  const int n = sizeof(h) / sizeof(h[0]);
  float *dx, *dout;
  hipMalloc(&dx, sizeof(h)); hipMalloc(&dout, 4 * n * 4);
  hipMemcpy(dx, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dx, dout, n);
  float o[128]; hipMemcpy(o, dout, 4 * n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("x=%g: exp2(x)=%g exp2(fma(x,s,-x s))=%g exp2(-inf-x)=%g fma(x,s,inf)=%g\n", h[i], o[i], o[n + i], o[2 * n + i], o[3 * n + i]);
  return 0;
}
