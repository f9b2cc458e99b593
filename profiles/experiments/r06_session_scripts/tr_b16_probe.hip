// Probe of ds_read_b64_tr_b16 on gfx950: every lane supplies the address of its own 8-byte chunk (4 halfs); which
// halfs does it get back?    hipcc --offload-arch=gfx950 tools/tr_b16_probe.hip -o tools/tr_b16_probe && tools/tr_b16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out, int pitch_halfs) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (_Float16)(float)(i % 2048);
  __syncthreads();
  const int l = threadIdx.x, li = l & 15, lg = l >> 4;
  // lane supplies: row (key) = 4*lg + (li >> 2), cols 4*(li & 3) .. +3 of a row-major [key][pitch] image
  const int idx = (4 * lg + (li >> 2)) * pitch_halfs + 4 * (li & 3);
  const unsigned addr = (unsigned)(size_t)(&lds[idx]);   // LDS byte address (low 32 bits of the generic pointer)
  half4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (float)v[j];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4);
  for (int pitch : {16, 80}) {
    probe<<<1, 64>>>(d, pitch);
    float h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("pitch %d halfs: lane -> 4 half indices (expect lane (li,lg) elem j = (4*lg + j)*pitch + li)\n", pitch);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
      printf("  l%2d:", l);
      for (int j = 0; j < 4; ++j) { printf(" %4.0f", h[l * 4 + j]); if ((int)h[l * 4 + j] != (4 * (l >> 4) + j) * pitch + (l & 15)) ++bad; }
      if (l % 4 == 3) printf("\n");
    }
    printf("mismatches vs expectation: %d\n", bad);
  }
  return 0;
}
