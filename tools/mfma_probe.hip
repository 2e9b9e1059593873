// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands and unit scales: which k does byte j of lane (r, h) hold?
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o variants/mfma_probe && variants/mfma_probe     (development tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// e4m3fn encodings of small non-negative integers 0..15 (exact)
__host__ __device__ inline uint8_t e4m3_of_int(int v) {
  static const uint8_t t[16] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E, 0x50, 0x51, 0x52, 0x53, 0x54, 0x55, 0x56, 0x57};
  return t[v];
}

__global__ void probe(const uint8_t* A /*[32][64]*/, const uint8_t* B /*[64][32]*/, float* C /*[32][32]*/, int scale) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  i32x8 a, b;
  uint8_t ab[32], bb[32];
  for (int j = 0; j < 32; ++j) { ab[j] = A[r * 64 + 32 * h + j]; bb[j] = B[(32 * h + j) * 32 + r]; }
  for (int w = 0; w < 8; ++w) {
    a[w] = ab[4 * w] | (ab[4 * w + 1] << 8) | (ab[4 * w + 2] << 16) | (ab[4 * w + 3] << 24);
    b[w] = bb[4 * w] | (bb[4 * w + 1] << 8) | (bb[4 * w + 2] << 16) | (bb[4 * w + 3] << 24);
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale, 0, scale);
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    C[row * 32 + r] = c[i];
  }
}

int main() {
  std::vector<uint8_t> A(32 * 64), B(64 * 32);
  std::vector<int> Ai(32 * 64), Bi(64 * 32);
  unsigned s = 12345;
  for (auto& v : Ai) { s = s * 1664525u + 1013904223u; v = (s >> 24) & 15; }
  for (auto& v : Bi) { s = s * 1664525u + 1013904223u; v = (s >> 24) & 15; }
  for (int i = 0; i < 32 * 64; ++i) { A[i] = e4m3_of_int(Ai[i]); B[i] = e4m3_of_int(Bi[i]); }
  uint8_t *dA, *dB; float* dC;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, 32 * 32 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  for (int scale : {0x7F7F7F7F, 0x7F, (int)0x80808080u, 0}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, scale);
    std::vector<float> C(32 * 32);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, ratio = 0; int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      int ref = 0; for (int k = 0; k < 64; ++k) ref += Ai[i * 64 + k] * Bi[k * 32 + j];
      double e = fabs(C[i * 32 + j] - ref); if (e > maxerr) maxerr = e; if (e > 0) bad++;
      if (ref) ratio = C[i * 32 + j] / ref;
    }
    printf("scale 0x%08x: max |C - ref| = %g, %d wrong, C/ref(last) = %g\n", scale, maxerr, bad, ratio);
  }
  return 0;
}
