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
