// gam_convmod.h -- the element-wise middle of the Conformer convolution module
// (reference gigaam/encoder.py:396-409), one fused HBM-bound kernel between the two
// pointwise-conv GEMMs:
//     GLU over channels -> zero padded frames -> depthwise Conv1d(k, groups=d) + bias
//     -> BatchNorm1d(eval, running stats) [v1/v2]  or  LayerNorm over channels [v3]
//     -> SiLU
// Algorithmic traffic: read [N, 2d] + write [N, d] fp32 (147.7 MB per layer at
// N = 16 032, d = 768).
#pragma once
#include "gam_common.h"
#include "gam_pack.h"
#include <type_traits>

struct GamConvModArgs {
  const float* u;      // [B*Ta, 2d]  pointwise_conv1 output (bias included)
  float* z;            // [B*Ta, d]
  const float* dw_w;   // [d, ks]
  const float* dw_b;   // [d]
  const float* n_scale;  // BN: gamma/sqrt(var+eps)   | LN: weight
  const float* n_shift;  // BN: beta - mean*scale     | LN: bias
  const int* lens;     // valid frames per utterance
  const int* cu;       // packed rows (gam_pack.h): first row of every utterance; null = padded layout (b * Ta)
  int B, Ta, Tv, d, ks;
  float eps;
  int z_split;   // z in the sp32 GEMM-operand layout
  int* range_flag;   // z_split: values beyond fp16's range set it (gam_common.h gam_range_note); may be null
};

// ---- BatchNorm variant: block = 64 channels x 128 frames, a thread owns 4 consecutive channels ----
// HBM-bound (read [N,2d] + write [N,d]).  Round 1 ran at 3.2 TB/s with 4-byte loads, 2-byte sp32 stores and
// 64-frame tiles (47 % halo at k = 31); here every global access is 16 bytes (a lane's 4 channels: f32x4 loads,
// one 16-byte fp32 store or two 8-byte sp32 stores), the tile is 128 frames (23 % halo, L2-served), and all
// ~20 loads of a thread are in flight before the first is used.  The depthwise taps run over a register window
// (8 outputs read 8 + KS - 1 tile rows once); the KS x 64 weights sit in LDS as [k][channel] and are read
// once per tap.  Per output the fmaf chain runs k = 0 .. KS-1 as before: bit-identical results.
template <int KS>
__global__ __launch_bounds__(512, 4) void gam_convmod_bn_kernel(GamConvModArgs a) {
  // 512 threads = 16 channel quads x 32 row groups, 4 outputs per thread: the register window stays at ~70 VGPRs, so
  // two workgroups (16 waves) share a CU -- with 8 outputs per thread (193 VGPRs, 8 waves per CU) the kernel ran
  // no faster than the 4-byte-load version it replaced (49 vs 45 us): too few waves to cover the load phase.
  constexpr int NRG = 32, TT = 128, PAD = (KS - 1) / 2, ROWS = (TT + KS - 1 + NRG - 1) / NRG * NRG, OUT = TT / NRG;
  __shared__ f32x4 tile[ROWS * 16];   // [row][quad]: 256-byte rows; a 16-lane ds_read_b128 group covers one row
  __shared__ f32x4 wl[KS * 16];       // [k][quad]
  const int tid = threadIdx.x;
  const int q = tid & 15, rg = tid >> 4;
  const int b = blockIdx.z, c = blockIdx.y * 64 + q * 4, t0 = blockIdx.x * TT;
  int klen = a.lens[b];
  klen = klen < a.Tv ? klen : a.Tv;
  const GamRows ur = gam_rows(a.cu, b, a.Ta, klen);
  if (t0 >= ur.lim) return;      // a tile behind the utterance's last row (whole workgroup)
  const size_t rowbase = ur.base;
  const int rlim = ur.lim;
  // GLU'd input tile: row r of the tile is loaded by row group r % 16 (clamped row, value masked afterwards --
  // a per-element "if in range: load" keeps one load outstanding per thread)
  constexpr int NLD = ROWS / NRG;     // (ROWS is rounded up to whole row groups: no "row in range" branch anywhere)
  f32x4 ua[NLD], ub[NLD];
#pragma unroll
  for (int u = 0; u < NLD; ++u) {
    const int t = t0 - PAD + rg + NRG * u;
    const int tc = t < 0 ? 0 : (t < rlim ? t : rlim - 1);
    const float* up = a.u + (rowbase + tc) * (size_t)(2 * a.d);
    ua[u] = *reinterpret_cast<const f32x4*>(up + c);
    ub[u] = *reinterpret_cast<const f32x4*>(up + a.d + c);
  }
  {   // weights [d][KS] -> wl[k][quad]: this thread stages taps k = rg + NRG i of its 4 channels
#pragma unroll
    for (int i = 0; i < (KS + NRG - 1) / NRG; ++i) {
      const int k = rg + NRG * i;
      if (k < KS) wl[k * 16 + q] = (f32x4){a.dw_w[(size_t)c * KS + k], a.dw_w[(size_t)(c + 1) * KS + k],
                                           a.dw_w[(size_t)(c + 2) * KS + k], a.dw_w[(size_t)(c + 3) * KS + k]};
    }
  }
#pragma unroll
  for (int u = 0; u < NLD; ++u) {
    const int rr = rg + NRG * u;
    const int t = t0 - PAD + rr;
    // a true select (v_cndmask, no branch), like the reference's masked_fill: padded / don't-care frames enter the taps as
    // exact zeros even when the grow-only workspace holds inf or NaN there (0 * NaN would leak into the valid frames)
    const bool live = t >= 0 && t < klen;
    const f32x4 glu = (f32x4){ua[u].x * gam_sigmoid(ub[u].x), ua[u].y * gam_sigmoid(ub[u].y),
                              ua[u].z * gam_sigmoid(ub[u].z), ua[u].w * gam_sigmoid(ub[u].w)};
    tile[rr * 16 + q] = live ? glu : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  // The taps run over a register window of OUT + KS - 1 tile rows.  With all four channels of the lane in one
  // window (38 x f32x4 + accumulators) hipcc spilled 36-62 VGPRs into the tap loop (112 us instead of 45), so the
  // lane's channels go in two passes of a channel PAIR each (38 x 8-byte LDS reads, ~120 live VGPRs).
  // Taps: output i needs tile rows i .. i + KS - 1.  Holding that whole window in registers (38 x f32x4) made hipcc
  // spill 36-70 VGPRs into the tap loop, and it hoists every loop-invariant LDS read to the top whatever the source
  // order.  So the window ROLLS: tap k uses rows k .. k+OUT-1, row k+OUT+1 and the weights of tap k+2 are fetched
  // during tap k, and each of those reads takes its address through an empty asm that also consumes an accumulator of
  // tap k -- ALL of them: pinning one component made hipcc run that component's whole chain first and spill 222
  // VGPRs -- so it cannot be issued earlier.  ~6 rows live at a time: the kernel fits 128 VGPRs, 16 waves per CU.
  const f32x4* trow = tile + (rg * OUT) * 16 + q;
  const f32x4* wq = wl + q;
  int lofs = 0;   // always 0; laundered through the asm statements below
  f32x4 win[OUT + KS - 1];
#pragma unroll
  for (int r = 0; r < OUT + 1; ++r) win[r] = trow[r * 16];
  f32x4 acc[OUT];
#pragma unroll
  for (int i = 0; i < OUT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 w_cur = wq[0], w_nxt = wq[16];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    const f32x4 w = w_cur;
    w_cur = w_nxt;
#pragma unroll
    for (int i = 0; i < OUT; ++i) {
      acc[i].x = fmaf(w.x, win[i + k].x, acc[i].x);
      acc[i].y = fmaf(w.y, win[i + k].y, acc[i].y);
      acc[i].z = fmaf(w.z, win[i + k].z, acc[i].z);
      acc[i].w = fmaf(w.w, win[i + k].w, acc[i].w);
    }
    static_assert(OUT == 4, "the pin below names the 16 accumulator components of a tap");
    asm volatile("" : "+v"(lofs), "+v"(acc[0].x), "+v"(acc[0].y), "+v"(acc[0].z), "+v"(acc[0].w), "+v"(acc[1].x), "+v"(acc[1].y),
                 "+v"(acc[1].z), "+v"(acc[1].w), "+v"(acc[2].x), "+v"(acc[2].y), "+v"(acc[2].z), "+v"(acc[2].w), "+v"(acc[3].x),
                 "+v"(acc[3].y), "+v"(acc[3].z), "+v"(acc[3].w));
    if (k + 2 < KS) w_nxt = wq[(k + 2) * 16 + lofs];
    if (k + OUT + 1 < OUT + KS - 1) win[k + OUT + 1] = trow[(k + OUT + 1) * 16 + lofs];
  }
  const f32x4 bias = *reinterpret_cast<const f32x4*>(a.dw_b + c);
  const f32x4 sc = *reinterpret_cast<const f32x4*>(a.n_scale + c);
  const f32x4 sh = *reinterpret_cast<const f32x4*>(a.n_shift + c);
#pragma unroll
  for (int i = 0; i < OUT; ++i) {
    const int t = t0 + rg * OUT + i;
    const f32x4 y = (acc[i] + bias) * sc + sh;
    const float v0 = gam_silu(y.x), v1 = gam_silu(y.y), v2 = gam_silu(y.z), v3 = gam_silu(y.w);
    if (t < rlim) {
      gam_range_note(a.range_flag, v0, v1, v2, v3);   // z feeds the pointwise-conv2 GEMM unscaled
      gam_store4(a.z, (rowbase + t) * (size_t)a.d, c, v0, v1, v2, v3, a.z_split);
    }
  }
}

// ---- LayerNorm variant (v3): block = 8 frames x all channels (d <= 1024) ----
template <int KS>
__global__ __launch_bounds__(256) void gam_convmod_ln_kernel(GamConvModArgs a) {
  constexpr int TT = 8, PAD = (KS - 1) / 2, ROWS = TT + KS - 1, MAXC = 4;
  extern __shared__ __attribute__((aligned(16))) float gam_smem_cm[];
  float* tile = gam_smem_cm;                 // [ROWS][d]
  float* red = gam_smem_cm + ROWS * a.d;     // [4 waves][TT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, t0 = blockIdx.x * TT;
  int klen = a.lens[b];
  klen = klen < a.Tv ? klen : a.Tv;
  const GamRows ur = gam_rows(a.cu, b, a.Ta, klen);
  if (t0 >= ur.lim) return;      // a tile behind the utterance's last row (whole workgroup)
  const size_t rowbase = ur.base;
  const int rlim = ur.lim;
  // GLU'd input tile, 4 rows (up to 32 loads per thread) in flight at a time; rows / channels out of
  // range are clamped for the load and masked afterwards (no per-element branch around a load)
  for (int r0 = 0; r0 < ROWS; r0 += 4) {
    float ua[4][MAXC], ub[4][MAXC];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = t0 - PAD + r0 + q;
      const int tc = t < 0 ? 0 : (t < rlim ? t : rlim - 1);
      const float* up = a.u + (rowbase + tc) * (size_t)(2 * a.d);
#pragma unroll
      for (int ci = 0; ci < MAXC; ++ci) {
        const int c = tid + ci * 256 < a.d ? tid + ci * 256 : a.d - 1;
        ua[q][ci] = up[c];
        ub[q][ci] = up[a.d + c];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rr = r0 + q, t = t0 - PAD + rr;
      const bool ok = t >= 0 && t < klen;
#pragma unroll
      for (int ci = 0; ci < MAXC; ++ci) {
        const int c = tid + ci * 256;
        if (rr < ROWS && c < a.d) tile[rr * a.d + c] = ok ? ua[q][ci] * gam_sigmoid(ub[q][ci]) : 0.f;
      }
    }
  }
  __syncthreads();
  float y[MAXC][TT];
#pragma unroll
  for (int ci = 0; ci < MAXC; ++ci) {
    const int c = tid + ci * 256;
    if (c < a.d) {
      float w[KS];
#pragma unroll
      for (int k = 0; k < KS; ++k) w[k] = a.dw_w[(size_t)c * KS + k];
      const float bias = a.dw_b[c];
#pragma unroll
      for (int i = 0; i < TT; ++i) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < KS; ++k) acc = fmaf(w[k], tile[(i + k) * a.d + c], acc);
        y[ci][i] = acc + bias;
      }
    } else {
#pragma unroll
      for (int i = 0; i < TT; ++i) y[ci][i] = 0.f;
    }
  }
  // per-frame mean over channels
  float mean[TT], rstd[TT];
#pragma unroll
  for (int i = 0; i < TT; ++i) {
    float s = 0.f;
#pragma unroll
    for (int ci = 0; ci < MAXC; ++ci) s += y[ci][i];
    s = gam_wave_sum(s);
    if (lane == 0) red[wave * TT + i] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TT; ++i) mean[i] = (red[i] + red[TT + i] + red[2 * TT + i] + red[3 * TT + i]) / (float)a.d;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TT; ++i) {
    float s = 0.f;
#pragma unroll
    for (int ci = 0; ci < MAXC; ++ci) {
      const int c = tid + ci * 256;
      if (c < a.d) { const float dlt = y[ci][i] - mean[i]; s += dlt * dlt; }
    }
    s = gam_wave_sum(s);
    if (lane == 0) red[wave * TT + i] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TT; ++i)
    rstd[i] = 1.0f / sqrtf((red[i] + red[TT + i] + red[2 * TT + i] + red[3 * TT + i]) / (float)a.d + a.eps);
#pragma unroll
  for (int ci = 0; ci < MAXC; ++ci) {
    const int c = tid + ci * 256;
    if (c < a.d) {
      const float g = a.n_scale[c], be = a.n_shift[c];
#pragma unroll
      for (int i = 0; i < TT; ++i) {
        const int t = t0 + i;
        if (t < rlim) gam_store1(a.z, (rowbase + t) * (size_t)a.d, c, gam_silu((y[ci][i] - mean[i]) * rstd[i] * g + be), a.z_split);
      }
    }
  }
}

// ---- LayerNorm variant, small kernels (v3 models: k = 5): block = 16 frames x all channels, a thread owns 4
// consecutive channels.  The depthwise conv mixes time only, so the thread convolves its own channels straight from
// the rows it loaded (no LDS tile); only the per-frame mean / variance over channels crosses threads.  16-byte loads
// and stores, all (16 + KS - 1) x 2 loads of a thread in flight, 25 % halo instead of 50 %.
template <int KS>
__global__ __launch_bounds__(256) void gam_convmod_ln4_kernel(GamConvModArgs a) {
  constexpr int TT = 16, PAD = (KS - 1) / 2, ROWS = TT + KS - 1;
  __shared__ float red[4][TT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, t0 = blockIdx.x * TT;
  const bool live = tid * 4 < a.d;                 // d % 4 == 0, d <= 1024
  const int c = live ? tid * 4 : 0;
  int klen = a.lens[b];
  klen = klen < a.Tv ? klen : a.Tv;
  const GamRows ur = gam_rows(a.cu, b, a.Ta, klen);
  if (t0 >= ur.lim) return;      // a tile behind the utterance's last row (whole workgroup)
  const size_t rowbase = ur.base;
  const int rlim = ur.lim;
  f32x4 ua[ROWS], ub[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int t = t0 - PAD + r;
    const int tc = t < 0 ? 0 : (t < rlim ? t : rlim - 1);
    const float* up = a.u + (rowbase + tc) * (size_t)(2 * a.d);
    ua[r] = *reinterpret_cast<const f32x4*>(up + c);
    ub[r] = *reinterpret_cast<const f32x4*>(up + a.d + c);
  }
  f32x4 w[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k)
    w[k] = (f32x4){a.dw_w[(size_t)c * KS + k], a.dw_w[(size_t)(c + 1) * KS + k], a.dw_w[(size_t)(c + 2) * KS + k],
                   a.dw_w[(size_t)(c + 3) * KS + k]};
  const f32x4 bias = *reinterpret_cast<const f32x4*>(a.dw_b + c);
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {   // GLU, padded frames enter the taps as 0 (a select, not a branch)
    const int t = t0 - PAD + r;
    const bool live = t >= 0 && t < klen;   // (true select: 0 * NaN from a don't-care frame must not reach valid frames)
    const f32x4 glu = (f32x4){ua[r].x * gam_sigmoid(ub[r].x), ua[r].y * gam_sigmoid(ub[r].y), ua[r].z * gam_sigmoid(ub[r].z),
                              ua[r].w * gam_sigmoid(ub[r].w)};
    ua[r] = live ? glu : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  f32x4 y[TT];
#pragma unroll
  for (int i = 0; i < TT; ++i) {
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      acc.x = fmaf(w[k].x, ua[i + k].x, acc.x);
      acc.y = fmaf(w[k].y, ua[i + k].y, acc.y);
      acc.z = fmaf(w[k].z, ua[i + k].z, acc.z);
      acc.w = fmaf(w[k].w, ua[i + k].w, acc.w);
    }
    y[i] = acc + bias;
  }
  // per-frame mean, then variance, over the d channels (two passes like native_layer_norm)
  float mean[TT], rstd[TT];
#pragma unroll
  for (int i = 0; i < TT; ++i) {
    float s = live ? (y[i].x + y[i].y) + (y[i].z + y[i].w) : 0.f;
    s = gam_wave_sum(s);
    if (lane == 0) red[wave][i] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TT; ++i) mean[i] = (red[0][i] + red[1][i] + red[2][i] + red[3][i]) / (float)a.d;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TT; ++i) {
    const float d0 = y[i].x - mean[i], d1 = y[i].y - mean[i], d2 = y[i].z - mean[i], d3 = y[i].w - mean[i];
    float s = live ? (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) : 0.f;
    s = gam_wave_sum(s);
    if (lane == 0) red[wave][i] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TT; ++i) rstd[i] = 1.0f / sqrtf((red[0][i] + red[1][i] + red[2][i] + red[3][i]) / (float)a.d + a.eps);
  if (!live) return;
  const f32x4 gm = *reinterpret_cast<const f32x4*>(a.n_scale + c);
  const f32x4 be = *reinterpret_cast<const f32x4*>(a.n_shift + c);
#pragma unroll
  for (int i = 0; i < TT; ++i) {
    const int t = t0 + i;
    if (t < rlim) {
      const float v0 = gam_silu((y[i].x - mean[i]) * rstd[i] * gm.x + be.x), v1 = gam_silu((y[i].y - mean[i]) * rstd[i] * gm.y + be.y);
      const float v2 = gam_silu((y[i].z - mean[i]) * rstd[i] * gm.z + be.z), v3 = gam_silu((y[i].w - mean[i]) * rstd[i] * gm.w + be.w);
      gam_range_note(a.range_flag, v0, v1, v2, v3);   // z feeds the pointwise-conv2 GEMM unscaled
      gam_store4(a.z, (rowbase + t) * (size_t)a.d, c, v0, v1, v2, v3, a.z_split);
    }
  }
}

static inline hipError_t gam_launch_convmod(const GamConvModArgs& a, int layer_norm, hipStream_t s) {
  if (!layer_norm) {
    if (a.d % 64 != 0) return hipErrorInvalidValue;
    dim3 grid(gam_cdiv(a.Ta, 128), a.d / 64, a.B);
    if (a.ks == 31) hipLaunchKernelGGL(gam_convmod_bn_kernel<31>, grid, dim3(512), 0, s, a);
    else if (a.ks == 5) hipLaunchKernelGGL(gam_convmod_bn_kernel<5>, grid, dim3(512), 0, s, a);
    else if (a.ks == 9) hipLaunchKernelGGL(gam_convmod_bn_kernel<9>, grid, dim3(512), 0, s, a);
    else return hipErrorInvalidValue;
  } else {
    if (a.d > 1024) return hipErrorInvalidValue;
    dim3 grid(gam_cdiv(a.Ta, 8), a.B);
    if ((a.ks == 5 || a.ks == 9) && a.d % 4 == 0) {   // thread-owns-4-channels kernel, 16-frame blocks
      dim3 g4(gam_cdiv(a.Ta, 16), a.B);
      if (a.ks == 5) hipLaunchKernelGGL(gam_convmod_ln4_kernel<5>, g4, dim3(256), 0, s, a);
      else hipLaunchKernelGGL(gam_convmod_ln4_kernel<9>, g4, dim3(256), 0, s, a);
    } else if (a.ks == 5) {
      const size_t sm = ((8 + 4) * a.d + 32) * sizeof(float);
      hipLaunchKernelGGL(gam_convmod_ln_kernel<5>, grid, dim3(256), sm, s, a);
    } else if (a.ks == 31) {
      const size_t sm = ((8 + 30) * a.d + 32) * sizeof(float);
      static std::atomic<unsigned long long> attr_devs{0};
      if (hipError_t e = gam_set_max_lds(reinterpret_cast<const void*>(gam_convmod_ln_kernel<31>), 160 * 1024, attr_devs)) return e;
      hipLaunchKernelGGL(gam_convmod_ln_kernel<31>, grid, dim3(256), sm, s, a);
    } else if (a.ks == 9) {
      const size_t sm = ((8 + 8) * a.d + 32) * sizeof(float);
      hipLaunchKernelGGL(gam_convmod_ln_kernel<9>, grid, dim3(256), sm, s, a);
    } else return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
