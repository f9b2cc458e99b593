// lds_canary.hip (r05 diagnosis) -- does some OTHER kernel write into this workgroup's LDS?
// Every workgroup fills `words` 4-byte words of dynamic LDS with a pattern of (address, workgroup), then re-reads all of it for
// `spin_us` microseconds and records every word that changed: count, first / last changed byte offset, a few (offset, value) samples.
// Nothing in this kernel writes LDS after the fill, so any change was made by someone else.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/lds_canary.hip -o tools/liblds_canary.so
#include <hip/hip_runtime.h>
#include <cstdint>

struct CanaryOut {            // per workgroup
  unsigned changed;           // words found changed (counted once per re-read pass that sees them)
  unsigned first_off, last_off;
  unsigned n_samples;
  unsigned sample_off[6], sample_val[6];
  unsigned xcc_cu;            // hardware id of the CU it ran on
  unsigned passes;
};

__device__ __forceinline__ unsigned pat(unsigned i, unsigned wg) { return 0xC0DE0000u ^ (i * 2654435761u) ^ (wg << 20); }

extern "C" __global__ __launch_bounds__(256) void lds_canary_kernel(CanaryOut* out, unsigned words, long long spin_ticks) {
  extern __shared__ unsigned canary_lds[];
  const unsigned wg = blockIdx.x, tid = threadIdx.x;
  for (unsigned i = tid; i < words; i += 256) canary_lds[i] = pat(i, wg);
  __shared__ unsigned s_changed, s_first, s_last, s_ns;
  if (tid == 0) { s_changed = 0; s_first = 0xffffffffu; s_last = 0; s_ns = 0; }
  __syncthreads();
  const long long t0 = wall_clock64();
  unsigned passes = 0;
  while (wall_clock64() - t0 < spin_ticks) {
    for (unsigned i = tid; i < words; i += 256) {
      const unsigned v = canary_lds[i];
      if (v != pat(i, wg)) {
        atomicAdd(&s_changed, 1u);
        atomicMin(&s_first, i * 4u);
        atomicMax(&s_last, i * 4u);
        const unsigned k = atomicAdd(&s_ns, 1u);
        if (k < 6) { out[wg].sample_off[k] = i * 4u; out[wg].sample_val[k] = v; }
        canary_lds[i] = pat(i, wg);      // repair, so that a later hit on the same word is seen again
      }
    }
    ++passes;
    __builtin_amdgcn_s_sleep(8);
  }
  __syncthreads();
  if (tid == 0) {
    out[wg].changed = s_changed; out[wg].first_off = s_first; out[wg].last_off = s_last; out[wg].n_samples = s_ns < 6 ? s_ns : 6;
    unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[wg].xcc_cu = hw; out[wg].passes = passes;
  }
}

extern "C" int lds_canary_launch(void* out_dev, int n_wg, int lds_bytes, double spin_us, void* stream) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(lds_canary_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess) return -1;   // (+ 16 B of static LDS)
    attr = true;
  }
  hipLaunchKernelGGL(lds_canary_kernel, dim3(n_wg), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, reinterpret_cast<CanaryOut*>(out_dev),
                     (unsigned)(lds_bytes / 4), (long long)(spin_us * 100.0));
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- canary 2: are this workgroup's VGPR-returning GLOBAL LOADS reliable while a neighbour streams LDS-DMA?
// Every thread re-reads 16-byte pieces of a read-only buffer whose words are a function of their address, with `inflight` loads outstanding
// at a time (like the decode kernel's weight loads), for spin_ticks; counts wrong words.  The LDS allocation only makes it a co-resident-able
// workgroup of the decode's size.
struct Canary2Out { unsigned bad, first_word, got, want, passes, pad[3]; };
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pat2(unsigned w) { return 0xA5000000u ^ (w * 2246822519u); }

extern "C" __global__ __launch_bounds__(256) void load_canary_kernel(Canary2Out* out, const unsigned* __restrict__ buf, unsigned n_words, long long spin_ticks) {
  extern __shared__ unsigned canary_lds[];
  const unsigned wg = blockIdx.x, tid = threadIdx.x;
  canary_lds[tid] = tid;
  __shared__ unsigned s_bad;
  if (tid == 0) { s_bad = 0; out[wg].first_word = 0xffffffffu; }
  __syncthreads();
  const long long t0 = wall_clock64();
  unsigned passes = 0;
  const unsigned n4 = n_words / 4;
  unsigned base = (wg * 977u + tid) % n4;
  while (wall_clock64() - t0 < spin_ticks) {
    u32x4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const u32x4*>(buf + 4 * ((base + 256u * u) % n4));
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const unsigned w0 = 4 * ((base + 256u * u) % n4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (v[u][e] != pat2(w0 + e)) {
          if (atomicAdd(&s_bad, 1u) == 0) { out[wg].first_word = w0 + e; out[wg].got = v[u][e]; out[wg].want = pat2(w0 + e); }
        }
    }
    base = (base + 4099u) % n4;
    ++passes;
  }
  __syncthreads();
  if (tid == 0) { out[wg].bad = s_bad; out[wg].passes = passes; }
}

extern "C" __global__ void load_canary_fill(unsigned* buf, unsigned n_words) {
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += gridDim.x * blockDim.x) buf[i] = pat2(i);
}

extern "C" int load_canary_launch(void* out_dev, void* buf, unsigned n_words, int fill, int n_wg, int lds_bytes, double spin_us, void* stream) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(load_canary_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess) return -1;
    attr = true;
  }
  if (fill) hipLaunchKernelGGL(load_canary_fill, dim3(1024), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<unsigned*>(buf), n_words);
  else hipLaunchKernelGGL(load_canary_kernel, dim3(n_wg), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, reinterpret_cast<Canary2Out*>(out_dev),
                          reinterpret_cast<const unsigned*>(buf), n_words, (long long)(spin_us * 100.0));
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
