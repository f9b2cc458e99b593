// pkfma_canary.hip (r06 diagnosis) -- is v_pk_fma_f32 reliable in a wave that shares its SIMD with another kernel's MFMA-heavy waves?
//
// The audit build of the RNN-T cluster decode (gam_decode_cluster.h, -DGAM_RC_AUDIT=1) showed: beside the small-tile GEMM, a gate row's sum
// w . h recomputed by the SAME thread from the SAME operands (hashes equal) differs from the first evaluation -- always in groups of 8 / 16
// consecutive lanes, and ONLY in the row slots that hipcc had packed into the LOW half of a v_pk_fma_f32 (slots 0 and 2 of five; 0 of three);
// the scalar v_fmac_f32 slot and the high halves never.  Alone on the GPU: never.
//
// Every lane runs the same chain of fused multiply-adds twice per iteration -- once as scalar v_fmac_f32, once as v_pk_fma_f32 (both through
// inline asm, so the compiler cannot re-pack them) -- on operands held in registers (mode 0), re-read from LDS as one broadcast ds_read_b128
// per step (mode 1), or with the low / high halves of the packed operands assembled by v_mov_b32 right in front of the packed instruction as
// hipcc does it in the decode kernel (mode 2).  Any bit difference between the two evaluations is counted per half and per 8-lane group.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/pkfma_canary.hip -o tools/libpkfma_canary.so
#include <hip/hip_runtime.h>
#include <cstdint>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PkOut {               // per workgroup, 32 words
  unsigned bad_lo, bad_hi, iters, hw_id;
  unsigned group_hist[8];    // mismatches by (lane >> 3)
  unsigned n_samples, pad[3];
  unsigned s_lane[4], s_scalar[4], s_packed[4], s_iter[4];
};

#define PK_K 32   // chain length (register operands: 2 x 32 weights + 32 h values per lane)

__device__ __forceinline__ float pk_val(unsigned x) {   // a float in (-1, 1) from a hash
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return (float)(int)(x & 0xffffu) * (1.0f / 32768.0f) - 1.0f;
}

template <int MODE>
__global__ __launch_bounds__(256) void pkfma_canary_kernel(PkOut* out, long long spin_ticks, float* gw) {
  extern __shared__ __attribute__((aligned(16))) float pk_lds[];
  const unsigned wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  __shared__ unsigned s_lo, s_hi, s_ns, s_hist[8];
  if (tid < 8) s_hist[tid] = 0;
  if (tid == 0) { s_lo = 0; s_hi = 0; s_ns = 0; }
  float w0[PK_K], w1[PK_K], h[PK_K];
#pragma unroll
  for (int k = 0; k < PK_K; ++k) {
    w0[k] = pk_val(tid * 131u + k * 7u + 1u) * 0.07f;
    w1[k] = pk_val(tid * 257u + k * 13u + 5u) * 0.07f;
    h[k] = pk_val(k * 31u + 3u) * 0.5f;            // the same for every lane, like the decode's broadcast h
  }
  for (int k = tid; k < PK_K; k += 256) pk_lds[k] = h[k];
  if constexpr (MODE == 4) {      // the same per-lane weight quads in GLOBAL memory (every workgroup writes the same values)
#pragma unroll
    for (int k = 0; k < PK_K; ++k) {
      gw[(((k >> 2) * 256 + tid) * 2 + 0) * 4 + (k & 3)] = w0[k];
      gw[(((k >> 2) * 256 + tid) * 2 + 1) * 4 + (k & 3)] = w1[k];
    }
    __threadfence();
  }
  if constexpr (MODE == 3) {
    float* wl = pk_lds + 64;
#pragma unroll
    for (int k = 0; k < PK_K; ++k) {
      wl[(((k >> 2) * 256 + tid) * 2 + 0) * 4 + (k & 3)] = w0[k];
      wl[(((k >> 2) * 256 + tid) * 2 + 1) * 4 + (k & 3)] = w1[k];
    }
  }
  __syncthreads();
  const long long t0 = wall_clock64();
  unsigned it = 0;
  while (wall_clock64() - t0 < spin_ticks) {
    const float seed0 = pk_val(it * 977u + tid), seed1 = pk_val(it * 613u + tid + 77u);
    float a0 = seed0, a1 = seed1;
    f32x2 p = (f32x2){seed0, seed1};
#pragma unroll
    for (int k = 0; k < PK_K; k += 4) {
      f32x4 hv;
      if constexpr (MODE >= 1) hv = *reinterpret_cast<const f32x4*>(pk_lds + k);   // broadcast read, as h_s in the decode
      else hv = (f32x4){h[k], h[k + 1], h[k + 2], h[k + 3]};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float hk = hv[e];
        asm("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(w0[k + e]), "v"(hk));
        asm("v_fmac_f32 %0, %1, %2" : "+v"(a1) : "v"(w1[k + e]), "v"(hk));
      }
      if constexpr (MODE == 2) {
        // the decode kernel's device code: the weight pair is assembled by two v_mov_b32 right in front of the packed FMA, h.x / h.y are
        // selected from ONE 64-bit register pair by op_sel
        f32x2 wp, hxy = (f32x2){hv.x, hv.y}, hzw = (f32x2){hv.z, hv.w};
        asm volatile("v_mov_b32 %0, %1" : "=v"(wp.y) : "v"(w1[k]));
        asm volatile("v_mov_b32 %0, %1" : "=v"(wp.x) : "v"(w0[k]));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p) : "v"(wp), "v"(hxy));
        asm volatile("v_mov_b32 %0, %1" : "=v"(wp.x) : "v"(w0[k + 1]));
        asm volatile("v_mov_b32 %0, %1" : "=v"(wp.y) : "v"(w1[k + 1]));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(p) : "v"(wp), "v"(hxy));
        asm volatile("v_mov_b32 %0, %1" : "=v"(wp.x) : "v"(w0[k + 2]));
        asm volatile("v_mov_b32 %0, %1" : "=v"(wp.y) : "v"(w1[k + 2]));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p) : "v"(wp), "v"(hzw));
        asm volatile("v_mov_b32 %0, %1" : "=v"(wp.x) : "v"(w0[k + 3]));
        asm volatile("v_mov_b32 %0, %1" : "=v"(wp.y) : "v"(w1[k + 3]));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(p) : "v"(wp), "v"(hzw));
      } else if constexpr (MODE != 3 && MODE != 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x2 wp = (f32x2){w0[k + e], w1[k + e]}, hp = (f32x2){hv[e], hv[e]};
          asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p) : "v"(wp), "v"(hp));
        }
      }
    }
    if constexpr (MODE == 3) {
      // hipcc's own packing, as in the decode kernel: two rows' weights arrive as 16-byte quads (here from LDS, per lane), the pair
      // (row0.k, row1.k) is assembled by v_mov_b32 in front of each v_pk_fma_f32 and h.k is selected by op_sel from the broadcast quad
      p = (f32x2){seed0, seed1};
      const float* wl = pk_lds + 64;
#pragma unroll
      for (int k = 0; k < PK_K; k += 4) {
        const f32x4 hv = *reinterpret_cast<const f32x4*>(pk_lds + k);
        const f32x4 wa = *reinterpret_cast<const f32x4*>(wl + (((k >> 2) * 256 + tid) * 2 + 0) * 4);
        const f32x4 wb = *reinterpret_cast<const f32x4*>(wl + (((k >> 2) * 256 + tid) * 2 + 1) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) p = __builtin_elementwise_fma((f32x2){wa[e], wb[e]}, (f32x2){hv[e], hv[e]}, p);
      }
    }
    if constexpr (MODE == 4) {
      // as mode 3 with the weight quads loaded from global memory, all sixteen in flight, consumed behind hipcc's counted vmcnt waits: the
      // decode kernel's gate loop in miniature
      p = (f32x2){seed0, seed1};
      f32x4 wa[PK_K / 4], wb[PK_K / 4];
#pragma unroll
      for (int k4 = 0; k4 < PK_K / 4; ++k4) {
        wa[k4] = *reinterpret_cast<const f32x4*>(gw + ((k4 * 256 + tid) * 2 + 0) * 4);
        wb[k4] = *reinterpret_cast<const f32x4*>(gw + ((k4 * 256 + tid) * 2 + 1) * 4);
      }
#pragma unroll
      for (int k4 = 0; k4 < PK_K / 4; ++k4) {
        const f32x4 hv = *reinterpret_cast<const f32x4*>(pk_lds + 4 * k4);
#pragma unroll
        for (int e = 0; e < 4; ++e) p = __builtin_elementwise_fma((f32x2){wa[k4][e], wb[k4][e]}, (f32x2){hv[e], hv[e]}, p);
      }
    }
    const bool blo = __float_as_uint(p.x) != __float_as_uint(a0), bhi = __float_as_uint(p.y) != __float_as_uint(a1);
    if (blo || bhi) {
      if (blo) atomicAdd(&s_lo, 1u);
      if (bhi) atomicAdd(&s_hi, 1u);
      atomicAdd(&s_hist[lane >> 3], 1u);
      const unsigned kx = atomicAdd(&s_ns, 1u);
      if (kx < 4) {
        out[wg].s_lane[kx] = tid | (blo ? 0x10000u : 0u) | (bhi ? 0x20000u : 0u);
        out[wg].s_scalar[kx] = __float_as_uint(blo ? a0 : a1); out[wg].s_packed[kx] = __float_as_uint(blo ? p.x : p.y); out[wg].s_iter[kx] = it;
      }
    }
    ++it;
  }
  __syncthreads();
  if (tid == 0) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[wg].bad_lo = s_lo; out[wg].bad_hi = s_hi; out[wg].iters = it; out[wg].hw_id = hw; out[wg].n_samples = s_ns < 4 ? s_ns : 4;
    for (int i = 0; i < 8; ++i) out[wg].group_hist[i] = s_hist[i];
  }
}

extern "C" int pkfma_canary_launch(void* out_dev, int mode, int n_wg, int lds_bytes, double spin_us, void* stream, void* gw) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(pkfma_canary_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(pkfma_canary_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(pkfma_canary_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(pkfma_canary_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(pkfma_canary_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess)
      return -1;
    attr = true;
  }
  PkOut* o = reinterpret_cast<PkOut*>(out_dev);
  const long long ticks = (long long)(spin_us * 100.0);
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0) hipLaunchKernelGGL(pkfma_canary_kernel<0>, dim3(n_wg), dim3(256), (size_t)lds_bytes, s, o, ticks, (float*)gw);
  else if (mode == 1) hipLaunchKernelGGL(pkfma_canary_kernel<1>, dim3(n_wg), dim3(256), (size_t)lds_bytes, s, o, ticks, (float*)gw);
  else if (mode == 2) hipLaunchKernelGGL(pkfma_canary_kernel<2>, dim3(n_wg), dim3(256), (size_t)lds_bytes, s, o, ticks, (float*)gw);
  else if (mode == 4) hipLaunchKernelGGL(pkfma_canary_kernel<4>, dim3(n_wg), dim3(256), (size_t)lds_bytes, s, o, ticks, (float*)gw);
  else {
    if (lds_bytes < 72 * 1024) lds_bytes = 72 * 1024;   // 256 B of h + 64 KB of per-lane weight quads
    hipLaunchKernelGGL(pkfma_canary_kernel<3>, dim3(n_wg), dim3(256), (size_t)lds_bytes, s, o, ticks, (float*)gw);
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
