// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands and unit scales: does byte j of lane (i, g) hold k = 32 g + j
// of row / column i, and does the result follow the 16x16 C layout (col = lane & 15, row = 4 (lane >> 4) + reg)?
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_probe16.hip -o variants/mfma_probe16 && variants/mfma_probe16   (development tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline uint8_t e4m3_of_int(int v) {  // e4m3fn encodings of 0..15 (exact)
  const uint8_t t[16] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E, 0x50, 0x51, 0x52, 0x53, 0x54, 0x55, 0x56, 0x57};
  return t[v];
}

__global__ void probe(const uint8_t* A /*[16][128]*/, const uint8_t* B /*[128][16]*/, float* C /*[16][16]*/, int scale) {
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  i32x8 a, b;
  for (int w = 0; w < 8; ++w) {
    unsigned aw = 0, bw = 0;
    for (int e = 0; e < 4; ++e) {
      aw |= (unsigned)A[i * 128 + 32 * g + 4 * w + e] << (8 * e);
      bw |= (unsigned)B[(32 * g + 4 * w + e) * 16 + i] << (8 * e);
    }
    a[w] = (int)aw;
    b[w] = (int)bw;
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scale, 0, scale);
  for (int r = 0; r < 4; ++r) C[(4 * g + r) * 16 + i] = c[r];
}

int main() {
  std::vector<uint8_t> A(16 * 128), B(128 * 16);
  std::vector<int> Ai(16 * 128), Bi(128 * 16);
  unsigned s = 12345;
  for (auto& v : Ai) { s = s * 1664525u + 1013904223u; v = (s >> 24) & 15; }
  for (auto& v : Bi) { s = s * 1664525u + 1013904223u; v = (s >> 24) & 15; }
  for (int i = 0; i < 16 * 128; ++i) { A[i] = e4m3_of_int(Ai[i]); B[i] = e4m3_of_int(Bi[i]); }
  uint8_t *dA, *dB; float* dC;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, 16 * 16 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  for (int scale : {0x7F7F7F7F, 0x7F}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, scale);
    std::vector<float> C(16 * 16);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, ratio = 0; int bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      int ref = 0; for (int k = 0; k < 128; ++k) ref += Ai[i * 128 + k] * Bi[k * 16 + j];
      double e = fabs(C[i * 16 + j] - ref); if (e > maxerr) maxerr = e; if (e > 0) bad++;
      if (ref) ratio = C[i * 16 + j] / ref;
    }
    printf("16x16x128 scale 0x%08x: max |C - ref| = %g, %d wrong, C/ref(last) = %g\n", scale, maxerr, bad, ratio);
  }
  return 0;
}
