// gam_common.h -- shared helpers for the gfx950 (MI355X) kernels of libgigaam_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#define GAM_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int gam_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float gam_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float gam_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// sigmoid / SiLU on the hardware transcendentals (v_exp_f32 = 2^x, v_rcp_f32; ~1 ulp each):
// 4 VALU instructions instead of ~40 for expf + IEEE divide -- the SiLU epilogue of the
// FFN-up GEMM was costing as much as its main loop.
__device__ __forceinline__ float gam_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float gam_silu(float x) { return x * gam_sigmoid(x); }
__device__ __forceinline__ float gam_sigmoid_exact(float x) { return 1.0f / (1.0f + expf(-x)); }

// activation ids shared by host and device
enum { GAM_ACT_NONE = 0, GAM_ACT_SILU = 1, GAM_ACT_RELU = 2 };
