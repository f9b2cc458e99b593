// gam_stem.h -- HBM-bound producers/consumers around the stem GEMMs
// (reference gigaam/encoder.py:32-130) and layout changes at the C-ABI boundary.
#pragma once
#include "gam_common.h"

// ---------------------------------------------------------------------------------
// lengths: feat_len (i64, caller) -> int32 per-stage lengths.  calc_output_length
// (encoder.py:77-90) with kernel k odd, padding (k-1)/2, stride 2 is
// floor((L-1)/2 + 1) = (L+1)/2 for L >= 0; applied once per conv stage.
//   len0 = clamp(feat_len, 0, T); len1, len2 for masks; enc_len from the raw length.
// ---------------------------------------------------------------------------------
__global__ void gam_lengths_kernel(const long long* feat_len, int B, int T, int stages,
                                   int* len0, int* len1, int* len2, int* enc_len) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  long long raw = feat_len[b];
  if (raw < 0) raw = 0;
  long long l = raw < T ? raw : T;
  len0[b] = (int)l;
  long long l1 = (l + 1) / 2, l2 = (l1 + 1) / 2;
  len1[b] = (int)l1;
  len2[b] = (int)l2;
  long long e = raw;
  for (int s = 0; s < stages; ++s) e = (e + 1) / 2;
  enc_len[b] = (int)e;
}

// ---------------------------------------------------------------------------------
// Conv2d #1 (1 -> C, 3x3, stride 2, pad 1) + time mask + ReLU (encoder.py:59-69,
// 117-123), written channels-last into the zero-bordered image the implicit-GEMM
// Conv2d #2 reads:  img[b][p][q][c],  p = t1 + 1 in [0, 2*Ta), q = f1 + 1 in [0, FP).
// One workgroup per (b, p) row; thread = channel(s), 9 weights in registers, the three
// input rows broadcast from LDS; every store is a coalesced run over channels.
// ---------------------------------------------------------------------------------
struct GamConv1Args {
  const float* feat;   // [B, F, T]
  float* img;          // [B, 2*Ta, FP, C]
  const float* w;      // [C, 9]
  const float* bias;   // [C]
  const int* len0;     // valid input frames
  const int* len1;     // valid output frames of this stage
  int B, T, F, Ta, FP, C, T1;
  int img_split;       // 1: channels of a pixel in the sp32 GEMM-operand layout (C % 32 == 0); 2: dense fp16 image (C % 64 == 0)
  int* range_flag;     // img_split: a pixel beyond fp16's range sets it (gam_common.h gam_range_note); may be null
};

// FMT = a.img_split as a compile-time constant: the per-element store of this write-bound kernel takes no format branch
// (with the run-time flag and its third case the kernel went from 588 to 643 us at 32 x 20 s)
template <int FMT>
__global__ __launch_bounds__(256) void gam_conv2d1_kernel(GamConv1Args a) {
  __shared__ float xin[3][132];   // f = -1 .. F (F <= 128)
  const int tid = threadIdx.x;
  const int p = blockIdx.x, b = blockIdx.y;
  const int t1 = p - 1;
  // (format 2 of gam_store*: a dense fp16 image -- the row starts at the same ELEMENT offset of the buffer viewed as halfs)
  const size_t pix0 = ((size_t)b * 2 * a.Ta + p) * (size_t)a.FP * a.C;
  float* out = FMT == 2 ? reinterpret_cast<float*>(reinterpret_cast<_Float16*>(a.img) + pix0) : a.img + pix0;
  const int row_floats = FMT == 2 ? a.FP * a.C / 2 : a.FP * a.C;
  const bool live = t1 >= 0 && t1 < a.T1 && t1 < a.len1[b];
  if (!live) {
    for (int i = tid; i < row_floats; i += 256) out[i] = 0.f;
    return;
  }
  const int l0 = a.len0[b];
  for (int i = tid; i < 3 * (a.F + 2); i += 256) {
    const int kh = i / (a.F + 2), ff = i - kh * (a.F + 2);
    const int t = 2 * t1 - 1 + kh, f = ff - 1;
    float v = 0.f;
    if (t >= 0 && t < a.T && t < l0 && f >= 0 && f < a.F) v = a.feat[((size_t)b * a.F + f) * a.T + t];
    xin[kh][ff] = v;
  }
  __syncthreads();
  for (int c = tid; c < a.C; c += 256) {
    float w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = a.w[c * 9 + k];
    const float bias = a.bias[c];
    gam_store1(out, 0, c, 0.f, FMT);  // q = 0 border (zero in every format)
    for (int q = 1; q < a.FP; ++q) {
      const int fb = 2 * (q - 1);  // xin column of kw = 0  (f = 2*f1 - 1 -> index f + 1)
      float acc = bias;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc = fmaf(w[kh * 3 + kw], xin[kh][fb + kw], acc);
      acc = fmaxf(acc, 0.f);
      if (FMT != 0) gam_range_note(a.range_flag, acc, 0.f, 0.f, 0.f);   // the stored (post-ReLU) image feeds Conv2d#2 unscaled
      gam_store1(out, (size_t)q * a.C, c, acc, FMT);
    }
  }
}

// ---------------------------------------------------------------------------------
// v3 stem input: feat [B,F,T] -> time-major, masked, zero-padded rows
// xin[b][pad + t][f] inside a per-utterance stride of `rows` rows (Conv1d k=5, p=2).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gam_feat_to_rows_kernel(const float* feat, float* xin, const int* len0,
                                                               int B, int F, int T, int rows, int pad) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, f0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int l0 = len0[b];
  for (int i = ty; i < 32; i += 8) {
    const int f = f0 + i, t = t0 + tx;
    float v = 0.f;
    if (f < F && t < T && t < l0) v = feat[((size_t)b * F + f) * T + t];
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, f = f0 + tx;
    if (t < T && f < F) xin[((size_t)b * rows + pad + t) * F + f] = tile[tx][i];
  }
}

// ---------------------------------------------------------------------------------
// batched 2-D transpose  in[b][r][c] (row stride ld_in) -> out[b][c][r] (row stride ld_out)
// used for encoded [B,Ta,D] -> [B,D,T'] (encoder.py:647) and back for the heads.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gam_transpose_kernel(const float* in, float* out, int R, int Cc,
                                                            size_t in_bstride, long ld_in,
                                                            size_t out_bstride, long ld_out) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* ip = in + (size_t)b * in_bstride;
  float* op = out + (size_t)b * out_bstride;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    if (r < R && c < Cc) tile[i][tx] = ip[(size_t)r * ld_in + c];
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (r < R && c < Cc) op[(size_t)c * ld_out + r] = tile[tx][i];
  }
}

static inline hipError_t gam_launch_transpose(const float* in, float* out, int B, int R, int Cc,
                                              size_t in_bstride, long ld_in, size_t out_bstride, long ld_out,
                                              hipStream_t s) {
  if (B <= 0 || R <= 0 || Cc <= 0) return hipSuccess;
  dim3 grid(gam_cdiv(Cc, 32), gam_cdiv(R, 32), B);
  hipLaunchKernelGGL(gam_transpose_kernel, grid, dim3(256), 0, s, in, out, R, Cc, in_bstride, ld_in, out_bstride, ld_out);
  return hipGetLastError();
}
