// Probe of v_permlane32_swap_b32 on gfx950: which halves trade places, scalar form and the in-place form on a 16-register
// accumulator vector that the GEMM epilogue uses.  hipcc --offload-arch=gfx950 tools/permlane_probe.hip -o /tmp/pp && /tmp/pp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(unsigned* out, float* out2) {
  const unsigned lane = threadIdx.x;
  unsigned a = lane, b = 100 + lane;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[lane] = r[0];
  out[64 + lane] = r[1];
  f32x16 v;
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = (float)(lane * 100 + i);
#pragma unroll
  for (int pq = 0; pq < 2; ++pq)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#ifdef USE_BUILTIN   // (hipcc 7.2: a SEQUENCE of the builtin on vector elements is miscompiled -- every element comes back as element 0)
      const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v[8 * pq + e]), __builtin_bit_cast(unsigned, v[8 * pq + 4 + e]), false, false);
      v[8 * pq + e] = __builtin_bit_cast(float, sw[0]);
      v[8 * pq + 4 + e] = __builtin_bit_cast(float, sw[1]);
#else
      float x = v[8 * pq + e], y = v[8 * pq + 4 + e];
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
      v[8 * pq + e] = x;
      v[8 * pq + 4 + e] = y;
#endif
    }
#pragma unroll
  for (int i = 0; i < 16; ++i) out2[lane * 16 + i] = v[i];
}
int main() {
  unsigned* d; unsigned h[128]; float* d2; static float h2[64 * 16];
  hipMalloc(&d, sizeof h); hipMalloc(&d2, sizeof h2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, d2);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  hipMemcpy(h2, d2, sizeof h2, hipMemcpyDeviceToHost);
  printf("r0: lanes 0,1,31,32,33,63 = %u %u %u %u %u %u\n", h[0], h[1], h[31], h[32], h[33], h[63]);
  printf("r1: lanes 0,1,31,32,33,63 = %u %u %u %u %u %u\n", h[64], h[65], h[95], h[96], h[97], h[127]);
  for (int l : {0, 1, 32, 33}) { printf("lane %2d:", l); for (int i = 0; i < 16; ++i) printf(" %.0f", h2[l * 16 + i]); printf("\n"); }
  return 0;
}
