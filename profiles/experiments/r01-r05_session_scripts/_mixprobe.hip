#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
__global__ void k(const float* x, unsigned* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  float a = x[2 * i], b = x[2 * i + 1];
  // reference
  _Float16 h0 = (_Float16)a, h1 = (_Float16)b;
  _Float16 l0 = (_Float16)(a - (float)h0), l1 = (_Float16)(b - (float)h1);
  unsigned short rh0, rh1, rl0, rl1;
  memcpy(&rh0, &h0, 2); memcpy(&rh1, &h1, 2); memcpy(&rl0, &l0, 2); memcpy(&rl1, &l1, 2);
  unsigned hi, lo; float la, lb;
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
  asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(la) : "v"(hi), "v"(a));
  asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hi), "v"(b));
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(la), "v"(lb));
  out[6 * i + 0] = (unsigned)rh0 | ((unsigned)rh1 << 16);
  out[6 * i + 1] = (unsigned)rl0 | ((unsigned)rl1 << 16);
  out[6 * i + 2] = hi; out[6 * i + 3] = lo;
  out[6 * i + 4] = __float_as_uint(la); out[6 * i + 5] = __float_as_uint(a - (float)h0);
}
int main() {
  const int n = 1 << 16;
  float* hx = new float[n];
  srand(1);
  for (int i = 0; i < n; ++i) {
    float m = (float)rand() / RAND_MAX * 2 - 1;
    int e = rand() % 40 - 24;
    hx[i] = ldexpf(m, e);
  }
  hx[0] = 0.f; hx[1] = -0.f; hx[2] = 1.f; hx[3] = -1.f; hx[4] = 65504.f; hx[5] = 1e-8f; hx[6] = 3.14159f; hx[7] = -2.71828f;
  float* dx; unsigned* dout; unsigned* ho = new unsigned[3 * n];
  hipMalloc(&dx, n * 4); hipMalloc(&dout, 3 * n * 4);
  hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
  k<<<n / 2 / 256, 256>>>(dx, dout, n);
  hipMemcpy(ho, dout, 3 * n * 4, hipMemcpyDeviceToHost);
  int badh = 0, badl = 0;
  for (int i = 0; i < n / 2; ++i) {
    if (ho[6 * i] != ho[6 * i + 2]) { if (badh < 5) printf("hi mismatch a=%g b=%g ref %08x got %08x\n", hx[2 * i], hx[2 * i + 1], ho[6 * i], ho[6 * i + 2]); ++badh; }
    if (ho[6 * i + 1] != ho[6 * i + 3]) { if (badl < 8) printf("lo mismatch a=%g b=%g ref %08x got %08x  la %08x ref(a-h0) %08x\n", hx[2 * i], hx[2 * i + 1], ho[6 * i + 1], ho[6 * i + 3], ho[6 * i + 4], ho[6 * i + 5]); ++badl; }
  }
  printf("pairs %d: hi mismatches %d, lo mismatches %d\n", n / 2, badh, badl);
  return 0;
}
