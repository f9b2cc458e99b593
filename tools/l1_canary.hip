// l1_canary.hip (r06 diagnosis) -- are VGPR-returning global loads that HIT IN THE CU's L1 reliable while a co-resident workgroup of ANOTHER
// kernel streams LDS-DMA (global_load_lds) through the same L1?
//
// r05 established that an RNN-T cluster-decode workgroup sharing its CU with the small-tile LDS-DMA GEMM comes out perturbed, that the GEMM does
// not write into foreign LDS, and that loads on a 16 MB buffer (= L1 misses) beside it are clean.  What correlates with the perturbation in r05's
// own table is whether the decode re-reads a SMALL global array every round (its W_out slice, 20 KB: L1 hits) -- W_out in LDS: 0-1 / 300,
// W_out from global: 28-153 / 300-600 at the same cluster size -- and everything that empties the L1 often (buffer_inv sc1 per poll) cures it.
//
// victim:    every workgroup re-reads a region of `region_words` words (4-64 KB) of a read-only patterned buffer with 16 x 16-byte loads in flight
//            per thread, checks every word, records wrong ones (index, got, want).  Load flavour selectable: plain / nt / sc1 / sc0 sc1.
// aggressor: workgroups that stream LDS-DMA (mode 0: from a large buffer, 1: from an L2-hot 1 MB window), plain VGPR loads (2), MFMAs (3),
//            LDS traffic (4), LDS-DMA + MFMA (5).
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/l1_canary.hip -o tools/libl1_canary.so
#include <hip/hip_runtime.h>
#include <cstdint>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__host__ __device__ __forceinline__ unsigned l1c_pat(unsigned w) { return 0xA5000000u ^ (w * 2246822519u); }

struct L1VictimOut {          // per workgroup, 32 words
  unsigned bad, passes, hw_id, xcc_id;
  unsigned n_samples, pad[3];
  unsigned s_idx[8], s_got[8], s_want[8];
};

template <int MODE>
__device__ __forceinline__ u32x4 l1c_ld16(const unsigned* p) {
  if constexpr (MODE == 0) return *reinterpret_cast<const u32x4*>(p);
  else if constexpr (MODE == 1) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  else {
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    unsigned long long a, b;
    if constexpr (MODE == 2) {
      a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return (u32x4){(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void l1c_victim_kernel(L1VictimOut* out, const unsigned* __restrict__ buf, unsigned region_words, unsigned n_regions,
                                                          long long spin_ticks) {
  extern __shared__ unsigned l1c_lds[];
  const unsigned wg = blockIdx.x, tid = threadIdx.x;
  l1c_lds[tid] = tid;      // (the LDS allocation only sizes the workgroup like a decode workgroup)
  __shared__ unsigned s_bad, s_ns;
  if (tid == 0) { s_bad = 0; s_ns = 0; }
  __syncthreads();
  const unsigned n4 = region_words / 4;                         // 16-byte pieces of my region
  const unsigned rbase = (wg % n_regions) * region_words;       // first word of my region
  const long long t0 = wall_clock64();
  unsigned passes = 0, rot = tid;
  while (wall_clock64() - t0 < spin_ticks) {
    u32x4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = l1c_ld16<MODE>(buf + rbase + 4 * ((rot + 256u * u) % n4));
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const unsigned w0 = rbase + 4 * ((rot + 256u * u) % n4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (v[u][e] != l1c_pat(w0 + e)) {
          atomicAdd(&s_bad, 1u);
          const unsigned k = atomicAdd(&s_ns, 1u);
          if (k < 8) { out[wg].s_idx[k] = w0 + e; out[wg].s_got[k] = v[u][e]; out[wg].s_want[k] = l1c_pat(w0 + e); }
        }
    }
    rot = (rot + 37u) % n4;
    ++passes;
  }
  __syncthreads();
  if (tid == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[wg].bad = s_bad; out[wg].passes = passes; out[wg].hw_id = hw; out[wg].xcc_id = xcc; out[wg].n_samples = s_ns < 8 ? s_ns : 8;
  }
}

struct L1AggrOut { unsigned hw_id, xcc_id, iters, sink; };

// mode 0: LDS-DMA stream over the whole source buffer   1: LDS-DMA from an L2-hot 1 MB window   2: plain 16-byte loads (VGPR return) over the buffer
// mode 3: fp16 MFMA loop   4: LDS write / read loop   5: modes 1 + 3 interleaved (GEMM-like)
__global__ __launch_bounds__(256) void l1c_aggressor_kernel(L1AggrOut* out, const float* __restrict__ src, unsigned long long n_words, int mode, long long spin_ticks) {
  extern __shared__ __attribute__((aligned(16))) float l1c_alds[];   // >= 32 KiB: 8 x 1 KiB DMA slots per wave
  const unsigned wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long gw = (unsigned long long)wg * 4 + wave, nw = (unsigned long long)gridDim.x * 4;
  const unsigned long long span = (mode == 1 || mode == 5) ? (1ull << 18) : n_words;   // words
  float* my = l1c_alds + wave * 8 * 256;
  float sink = 0.f;
  f32x16 acc = {0};
  f16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(0.001f * (lane + i)); fb[i] = (_Float16)(0.002f * (lane - i)); }
  for (unsigned i = tid; i < 8 * 1024; i += 256) l1c_alds[i] = 0.f;
  __syncthreads();
  const long long t0 = wall_clock64();
  unsigned it = 0;
  while (wall_clock64() - t0 < spin_ticks) {
    if (mode == 0 || mode == 1 || mode == 5) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const unsigned long long piece = ((unsigned long long)it * 8 + k) * nw + gw;          // 1 KiB (256 words) per wave-instruction
        const float* p = src + (piece * 256 + lane * 4) % span;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)(my + k * 256), 16, 0, 0);
      }
    }
    if (mode == 3 || mode == 5) {
#pragma unroll
      for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
    }
    if (mode == 2) {
      u32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const unsigned long long piece = ((unsigned long long)it * 8 + k) * nw + gw;
        v[k] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned*>(src) + (piece * 256 + lane * 4) % span);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) sink += __uint_as_float((v[k].x ^ v[k].y ^ v[k].z ^ v[k].w) & 0x007fffffu);
    }
    if (mode == 4) {
#pragma unroll
      for (int k = 0; k < 8; ++k) my[k * 256 + lane * 4] = sink + k;
#pragma unroll
      for (int k = 0; k < 8; ++k) sink += my[k * 256 + ((lane * 4 + 64) & 255)];
    }
    if (mode == 0 || mode == 1 || mode == 5) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      sink += my[(it & 7) * 256 + lane];
    }
    ++it;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 16; ++i) sink += acc[i];
  if (lane == 0 && wave == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[wg].hw_id = hw; out[wg].xcc_id = xcc; out[wg].iters = it; out[wg].sink = __float_as_uint(sink);
  }
}

__global__ void l1c_fill_kernel(unsigned* buf, unsigned long long n_words) {
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (unsigned long long)gridDim.x * blockDim.x)
    buf[i] = l1c_pat((unsigned)i);
}

static int l1c_attr(const void* f) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) == hipSuccess ? 0 : -1; }

extern "C" int l1c_fill(void* buf, unsigned long long n_words, void* stream) {
  hipLaunchKernelGGL(l1c_fill_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<unsigned*>(buf), n_words);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int l1c_victim_launch(void* out_dev, const void* buf, unsigned region_words, unsigned n_regions, int mode, int n_wg, int lds_bytes, double spin_us,
                                 void* stream) {
  static bool attr = false;
  if (!attr) {
    if (l1c_attr(reinterpret_cast<const void*>(l1c_victim_kernel<0>)) || l1c_attr(reinterpret_cast<const void*>(l1c_victim_kernel<1>)) ||
        l1c_attr(reinterpret_cast<const void*>(l1c_victim_kernel<2>)) || l1c_attr(reinterpret_cast<const void*>(l1c_victim_kernel<3>)))
      return -1;
    attr = true;
  }
  L1VictimOut* o = reinterpret_cast<L1VictimOut*>(out_dev);
  const unsigned* b = reinterpret_cast<const unsigned*>(buf);
  const long long ticks = (long long)(spin_us * 100.0);
  hipStream_t s = (hipStream_t)stream;
  switch (mode) {
    case 0: hipLaunchKernelGGL(l1c_victim_kernel<0>, dim3(n_wg), dim3(256), (size_t)lds_bytes, s, o, b, region_words, n_regions, ticks); break;
    case 1: hipLaunchKernelGGL(l1c_victim_kernel<1>, dim3(n_wg), dim3(256), (size_t)lds_bytes, s, o, b, region_words, n_regions, ticks); break;
    case 2: hipLaunchKernelGGL(l1c_victim_kernel<2>, dim3(n_wg), dim3(256), (size_t)lds_bytes, s, o, b, region_words, n_regions, ticks); break;
    default: hipLaunchKernelGGL(l1c_victim_kernel<3>, dim3(n_wg), dim3(256), (size_t)lds_bytes, s, o, b, region_words, n_regions, ticks); break;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int l1c_aggressor_launch(void* out_dev, const void* src, unsigned long long n_words, int mode, int n_wg, int lds_bytes, double spin_us, void* stream) {
  static bool attr = false;
  if (!attr) {
    if (l1c_attr(reinterpret_cast<const void*>(l1c_aggressor_kernel))) return -1;
    attr = true;
  }
  if (lds_bytes < 32 * 1024) return -3;
  hipLaunchKernelGGL(l1c_aggressor_kernel, dim3(n_wg), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, reinterpret_cast<L1AggrOut*>(out_dev),
                     reinterpret_cast<const float*>(src), n_words, mode, (long long)(spin_us * 100.0));
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
