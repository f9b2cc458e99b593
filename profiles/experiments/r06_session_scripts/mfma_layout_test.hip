// Empirical check of the operand layouts assumed by gam_attn16.h (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k32(const float* A, const float* B, float* D) {   // A[16][32], B[32][16] -> D[16][16]
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  h8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)A[i * 32 + 8 * g + e]; b[e] = (_Float16)B[(8 * g + e) * 16 + i]; }
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = c[r];
}
__global__ void k16(const float* A, const float* B, float* D) {   // A[16][16], B[16][16]
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  h4 a, b;
  for (int e = 0; e < 4; ++e) { a[e] = (_Float16)A[i * 16 + 4 * g + e]; b[e] = (_Float16)B[(4 * g + e) * 16 + i]; }
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = c[r];
}
int main() {
  float hA[16 * 32], hB[32 * 16], hD[256], *dA, *dB, *dD;
  for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5) * 0.5f; }
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  for (int t = 0; t < 2; ++t) {
    const int K = t == 0 ? 32 : 16;
    if (t == 0) hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    else hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      double r = 0; for (int k = 0; k < K; ++k) r += (double)hA[i * K + k] * hB[k * 16 + j];
      err = fmax(err, fabs(r - hD[i * 16 + j]));
    }
    printf("mfma 16x16x%d f16: max err %g\n", K, err);
  }
  return 0;
}
