// gam_pack.h -- variable-length batches without padded rows (r06).
//   reference: gigaam/utils.py:103-155 (the optional flash-attn varlen path runs attention over unpadded tokens only); the rest of the
//   reference's encoder computes every padded frame of a ragged batch (gigaam/encoder.py:605-647) and masks it.
//
// Between the stem and the output transpose every activation is token-major [rows, D].  The padded layout gives utterance b the rows
// b*Ta .. b*Ta + Ta - 1 whatever its length; a ragged batch (SURVEY 8d's linspace(10 s, 20 s, 32): 25 % padding) then spends a quarter of
// every GEMM, LayerNorm and conv-module launch on rows no decoder reads.  The PACKED layout gives utterance b the rows cu[b] .. cu[b] + len[b] - 1:
//   * GEMMs / LayerNorms simply run on fewer rows (a row is a row); RoPE takes its frame index from row_t[row];
//   * attention and the conv module address utterance b at cu[b] and treat every frame outside [0, len[b]) as absent (GamRows below);
//   * the stem still runs on the padded layout (its strided implicit-GEMM addressing wants an affine row -> (b, t) map); its output is
//     gathered into packed rows once (49 MB at 32 x 20 s: ~20 us), and the output transpose scatters packed rows back into the API's
//     [B, D, T'] with zeros behind every utterance's last frame.
// The host must know the packed row count to size grids and pick GEMM tilings: gam_encode_varlen takes the lengths as a HOST array too
// (an upper bound per utterance is enough: the index kernel reports a device length above the host's through the handle's flag word).
#pragma once
#include "gam_common.h"

// rows of utterance b as a kernel sees them: first row, and how many of its frames exist as rows (padded: all Ta; packed: the valid ones)
struct GamRows { size_t base; int lim; };
__device__ __forceinline__ GamRows gam_rows(const int* cu, int b, int Ta, int klen) {
  GamRows r;
  if (cu != nullptr) { r.base = (size_t)cu[b]; r.lim = klen; }
  else { r.base = (size_t)b * Ta; r.lim = Ta; }
  return r;
}

// cu[b] = sum of len[b'] for b' < b (cu[B] = packed row count); row_t[p] = frame index of packed row p; row_src[p] = its row in the padded
// layout.  One workgroup: B is a batch size (<= a few hundred), the prefix is a serial loop of one thread, the fill is parallel.
// host_rows: what the host sized the launches for -- fewer than the device's count sets bit 1 of *flag (the caller's lengths were too short).
__global__ __launch_bounds__(256) void gam_pack_index_kernel(const int* __restrict__ len, int B, int Ta, int host_rows, int* __restrict__ cu,
                                                             int* __restrict__ row_t, int* __restrict__ row_src, int* flag) {
  __shared__ int s_cu[1025];
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int b = 0; b < B; ++b) { s_cu[b] = acc; acc += len[b] < 0 ? 0 : (len[b] < Ta ? len[b] : Ta); }
    s_cu[B] = acc;
    if (acc > host_rows && flag != nullptr) atomicOr(flag, 2);
  }
  __syncthreads();
  for (int b = threadIdx.x; b <= B; b += 256) cu[b] = s_cu[b];
  for (int b = 0; b < B; ++b) {
    const int n = s_cu[b + 1] - s_cu[b];
    for (int t = threadIdx.x; t < n; t += 256) {
      const int p = s_cu[b] + t;
      if (p < host_rows) { row_t[p] = t; row_src[p] = b * Ta + t; }
    }
  }
  // rows the host sized for beyond the device's count (its lengths were an upper bound): harmless copies of row 0
  for (int p = s_cu[B] + threadIdx.x; p < host_rows; p += 256) { row_t[p] = 0; row_src[p] = 0; }
}

// dst[p][:] = src[row_src[p]][:], 16 bytes per lane (D % 4 == 0)
__global__ __launch_bounds__(256) void gam_gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ row_src, float* __restrict__ dst,
                                                              int rows, int D) {
  const int per = D >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)rows * per; i += (size_t)gridDim.x * 256) {
    const int p = (int)(i / per), c = (int)(i - (size_t)p * per) * 4;
    *reinterpret_cast<float4*>(dst + (size_t)p * D + c) = *reinterpret_cast<const float4*>(src + (size_t)row_src[p] * D + c);
  }
}

// packed token-major rows -> the API's channel-first encoded [B, D, Tv] (encoder.py:647), zeros behind each utterance's last frame
__global__ __launch_bounds__(256) void gam_unpack_transpose_kernel(const float* __restrict__ x, const int* __restrict__ cu, const int* __restrict__ len,
                                                                   float* __restrict__ out, int Tv, int D) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  int n = len[b];
  n = n < 0 ? 0 : (n < Tv ? n : Tv);
  const float* ip = x + (size_t)cu[b] * D;
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, c = c0 + tx;
    tile[i][tx] = (t < n && c < D) ? ip[(size_t)t * D + c] : 0.f;
  }
  __syncthreads();
  float* op = out + (size_t)b * D * Tv;
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, t = t0 + tx;
    if (t < Tv && c < D) op[(size_t)c * Tv + t] = tile[tx][i];
  }
}

// packed rows -> token-major [B, Tv, D] (gam_encode_ex's tokens_out), zeros behind each utterance's last frame
__global__ __launch_bounds__(256) void gam_unpack_rows_kernel(const float* __restrict__ x, const int* __restrict__ cu, const int* __restrict__ len,
                                                              float* __restrict__ out, int B, int Tv, int D) {
  const int per = D >> 2;
  const size_t total = (size_t)B * Tv * per;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % per) * 4;
    const size_t bt = i / per;
    const int t = (int)(bt % Tv), b = (int)(bt / Tv);
    int n = len[b];
    n = n < 0 ? 0 : (n < Tv ? n : Tv);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < n) v = *reinterpret_cast<const float4*>(x + ((size_t)cu[b] + t) * D + c);
    *reinterpret_cast<float4*>(out + bt * D + c) = v;
  }
}
