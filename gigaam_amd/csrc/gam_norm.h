// gam_norm.h -- LayerNorm over d_model (reference gigaam/encoder.py:447,449,455,469,471)
// with the two fusions the Conformer layer allows:
//   MODE 1: y = LN(x) and yr = RoPE(y)  -- the reference rotates the layer-normed input
//           BEFORE the q/k projections (encoder.py:244-256, utils.py:83-100), so the
//           rotated copy feeds the fused W_q|W_k GEMM and the plain copy feeds W_v.
//   MODE 2: x' = LN_out(x) of layer i and y = LN_ff1(x') of layer i+1 in one pass.
// HBM-bound: one wave per row, 16-byte loads, the row lives in registers, two-pass
// mean/variance (fp32, eps 1e-5) like the reference's native_layer_norm.
#pragma once
#include "gam_common.h"

#define GAM_LN_MAXJ 4  // d_model <= 1024
#define GAM_LN_EARLY_ROWS 1024

struct GamLnArgs {
  const float* x;
  float* out1;        // MODE 0: y ; MODE 1: y ; MODE 2: x'
  float* out2;        // MODE 1: rope(y) ; MODE 2: y
  const float* w1; const float* b1;
  const float* w2; const float* b2;   // MODE 2
  const float* rcos; const float* rsin;  // MODE 1: [Tmax, dk/2]
  int rows, d, ta, dk;
  float eps;
  int split1, split2;   // write out1 / out2 in the sp32 GEMM-operand layout (gam_common.h)
  // Per-row pre-scale of the GEMM operand this kernel produces (gam_row_scale): rs[row] = 2^-e is what the
  // consuming GEMM multiplies its accumulators by.  MODE 0: out1; MODE 1: out1 AND out2 (one scale for the
  // plain and the rotated copy); MODE 2: out2 (out1 is the residual stream, never scaled).  The operand is
  // stored multiplied by 2^e in either format (sp32 or fp32: these outputs feed GEMMs only).  null = no scaling.
  float* rs;
  int rope_rows;        // rows of rcos / rsin (pos_emb_max_len): the stride-padding rows of a row block clamp to it
  // Fused split-K reduce (small grids: gam_gemm_sp.h split-K): instead of reading the row from x, build it from the
  // producing GEMM's partial sums -- x_row = resid + alpha (sum_s part[s] + bias), summed in slice order like
  // gam_splitk_reduce_kernel -- and store it to x (the residual stream) before normalising it.  One pass and one launch
  // instead of reduce kernel + LayerNorm kernel.  part == nullptr: plain LayerNorm.
  const float* part;    // [nsplit][rows][d] partial sums
  int nsplit;
  const float* pbias;   // [d]          (both mandatory with part: the launcher refuses a null)
  const float* presid;  // [rows][d]
  float palpha;
  float* xstore;        // where the finished row goes (MODE 0 / 1; MODE 2 overwrites it with out1 anyway)
  const int* row_t;     // packed rows (gam_pack.h): the frame index of every row for the rotary table; null = padded layout (row % ta)
};

// max |v| of a row held as float4[GAM_LN_MAXJ] across a wave
__device__ __forceinline__ float gam_ln_rowmax(const float4 (&v)[GAM_LN_MAXJ], int d, int lane) {
  float m = 0.f;
#pragma unroll
  for (int j = 0; j < GAM_LN_MAXJ; ++j)
    if ((j * 64 + lane) * 4 < d) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[j].x), fabsf(v[j].y))), fmaxf(fabsf(v[j].z), fabsf(v[j].w)));
  return gam_wave_max(m);
}

// affine parameters of one LayerNorm as the lanes hold them (loaded up front: see the kernel)
struct GamLnWB { float4 w[GAM_LN_MAXJ], b[GAM_LN_MAXJ]; };

__device__ __forceinline__ void gam_ln_load_wb(GamLnWB& p, const float* w, const float* b, int d, int lane) {
#pragma unroll
  for (int j = 0; j < GAM_LN_MAXJ; ++j) {
    const int c = (j * 64 + lane) * 4;
    p.w[j] = c < d ? *reinterpret_cast<const float4*>(w + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    p.b[j] = c < d ? *reinterpret_cast<const float4*>(b + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// PRE: the parameters are already in registers (p); otherwise they are read from w / b where they are used
template <bool PRE>
__device__ __forceinline__ void gam_ln_row(float4 (&v)[GAM_LN_MAXJ], int d, int lane, float eps, const GamLnWB& p,
                                           const float* w, const float* b) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < GAM_LN_MAXJ; ++j)
    if ((j * 64 + lane) * 4 < d) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  const float mean = gam_wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < GAM_LN_MAXJ; ++j)
    if ((j * 64 + lane) * 4 < d) {
      const float a0 = v[j].x - mean, a1 = v[j].y - mean, a2 = v[j].z - mean, a3 = v[j].w - mean;
      q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
  const float rstd = 1.0f / sqrtf(gam_wave_sum(q) / (float)d + eps);
#pragma unroll
  for (int j = 0; j < GAM_LN_MAXJ; ++j) {
    const int c = (j * 64 + lane) * 4;
    if (c < d) {
      const float4 ww = PRE ? p.w[j] : *reinterpret_cast<const float4*>(w + c);
      const float4 bb = PRE ? p.b[j] : *reinterpret_cast<const float4*>(b + c);
      v[j].x = (v[j].x - mean) * rstd * ww.x + bb.x;
      v[j].y = (v[j].y - mean) * rstd * ww.y + bb.y;
      v[j].z = (v[j].z - mean) * rstd * ww.z + bb.z;
      v[j].w = (v[j].w - mean) * rstd * ww.w + bb.w;
    }
  }
}

// Row of the fused split-K reduce: x_row = resid + alpha (sum_s part[s] + bias), slices summed in order (bit-identical to
// gam_splitk_reduce_kernel).  Straight-line on purpose -- NS is a compile-time count, lanes past d read column 0 instead of
// being masked off, bias and residual are mandatory -- so that every load of the row (NS slices + bias + residual per
// float4) is in flight before the first wait: the per-float4 form with its own waits was 9 dependent L2 round trips.
// NS = 0: any slice count, one load at a time.
template <int NS>
__device__ __forceinline__ void gam_ln_part_row(const GamLnArgs& a, int row, int lane, float4 (&v)[GAM_LN_MAXJ]) {
  const size_t slice = (size_t)a.rows * a.d;
  f32x4 sm[GAM_LN_MAXJ];
  float4 bb[GAM_LN_MAXJ], rr[GAM_LN_MAXJ];
#pragma unroll
  for (int j = 0; j < GAM_LN_MAXJ; ++j) {
    const int c0 = (j * 64 + lane) * 4, c = c0 < a.d ? c0 : 0;
    const float* p = a.part + (size_t)row * a.d + c;
    if constexpr (NS > 0) {
      sm[j] = gam_sum_slices_n<NS>(p, slice);
    } else {
      sm[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int s_ = 0; s_ < a.nsplit; ++s_) sm[j] += *reinterpret_cast<const f32x4*>(p + s_ * slice);
    }
    bb[j] = *reinterpret_cast<const float4*>(a.pbias + c);
    rr[j] = *reinterpret_cast<const float4*>(a.presid + (size_t)row * a.d + c);
  }
#pragma unroll
  for (int j = 0; j < GAM_LN_MAXJ; ++j) {
    const bool in = (j * 64 + lane) * 4 < a.d;
    float4 acc = make_float4(sm[j].x + bb[j].x, sm[j].y + bb[j].y, sm[j].z + bb[j].z, sm[j].w + bb[j].w);
    acc.x = acc.x * a.palpha + rr[j].x; acc.y = acc.y * a.palpha + rr[j].y;
    acc.z = acc.z * a.palpha + rr[j].z; acc.w = acc.w * a.palpha + rr[j].w;
    v[j] = in ? acc : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// PART: the fused split-K reduce (a.part != nullptr) -- its own instantiation, so that the plain kernel keeps its 24 VGPRs
// (the unrolled slice loads of the fused one take ~100).
// EARLY: the latency-bound form for small row counts (see below; 70-130 VGPRs) -- the bandwidth-bound large-batch launches
// keep the lean kernel and its full occupancy.
template <int MODE, bool PART = false, bool EARLY = false>
__global__ __launch_bounds__(256) void gam_layernorm_kernel(GamLnArgs a) {
  __shared__ float rowbuf[MODE == 1 ? 4 * GAM_LN_MAXJ * 256 : 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row_raw = blockIdx.x * 4 + wave;
  const bool live = row_raw < a.rows;
  const int row = live ? row_raw : a.rows - 1;
  const float* xr = a.x + (size_t)row * a.d;
  // Everything the row will need is requested before the row itself: the affine parameters (both sets in MODE 2) and, in
  // MODE 1, the row's rotary cos / sin.  At small row counts the kernel is a chain of dependent L2 / HBM round trips
  // (row -> parameters -> rotary table), ~2 us each with nothing else on the CU to hide them; issued together they cost one.
  GamLnWB p1, p2;
  if (EARLY) gam_ln_load_wb(p1, a.w1, a.b1, a.d, lane);
  if (EARLY && MODE == 2) gam_ln_load_wb(p2, a.w2, a.b2, a.d, lane);
  float4 rc4[GAM_LN_MAXJ], rs4[GAM_LN_MAXJ];
  if (EARLY && MODE == 1) {
    int t = a.row_t != nullptr ? a.row_t[row] : row % a.ta;
    t = t < a.rope_rows ? t : a.rope_rows - 1;   // stride-padding rows past pos_emb_max_len are don't-care frames
    const int half = a.dk >> 1;
#pragma unroll
    for (int j = 0; j < GAM_LN_MAXJ; ++j) {
      const int c = (j * 64 + lane) * 4;
      const int i0 = c % a.dk, ib = i0 < half ? i0 : i0 - half;
      rc4[j] = c < a.d ? *reinterpret_cast<const float4*>(a.rcos + (size_t)t * half + ib) : make_float4(0.f, 0.f, 0.f, 0.f);
      rs4[j] = c < a.d ? *reinterpret_cast<const float4*>(a.rsin + (size_t)t * half + ib) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  float4 v[GAM_LN_MAXJ];
  if constexpr (PART) {   // fused split-K reduce: the row is finished here (see GamLnArgs)
    switch (a.nsplit) {   // (uniform; the slice counts the plan can choose for K = 768 / 3072 -- gam_gemm_sp_plan)
      case 2: gam_ln_part_row<2>(a, row, lane, v); break;
      case 3: gam_ln_part_row<3>(a, row, lane, v); break;
      case 4: gam_ln_part_row<4>(a, row, lane, v); break;
      case 6: gam_ln_part_row<6>(a, row, lane, v); break;
      case 8: gam_ln_part_row<8>(a, row, lane, v); break;
      default: gam_ln_part_row<0>(a, row, lane, v); break;
    }
    if (live && MODE != 2) {   // (after every load of the row: a store in between would order the loads behind it)
#pragma unroll
      for (int j = 0; j < GAM_LN_MAXJ; ++j) {
        const int c = (j * 64 + lane) * 4;
        if (c < a.d) *reinterpret_cast<float4*>(a.xstore + (size_t)row * a.d + c) = v[j];
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < GAM_LN_MAXJ; ++j) {
      const int c = (j * 64 + lane) * 4;
      v[j] = c < a.d ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  gam_ln_row<EARLY>(v, a.d, lane, a.eps, p1, a.w1, a.b1);
  float sc1 = 1.0f, sc2 = 1.0f;    // applied to the stored values of an sp32 output
  if (MODE != 2 && a.rs != nullptr) {
    float s_, inv_;
    gam_row_scale(gam_ln_rowmax(v, a.d, lane), s_, inv_);
    if (live && lane == 0) a.rs[row] = inv_;
    sc1 = s_;
    sc2 = s_;
  }
  if (live) {
#pragma unroll
    for (int j = 0; j < GAM_LN_MAXJ; ++j) {
      const int c = (j * 64 + lane) * 4;
      if (c < a.d) gam_store4(a.out1, (size_t)row * a.d, c, v[j].x * sc1, v[j].y * sc1, v[j].z * sc1, v[j].w * sc1, a.split1);
    }
  }
  if (MODE == 2) {
    gam_ln_row<EARLY>(v, a.d, lane, a.eps, p2, a.w2, a.b2);
    if (a.rs != nullptr) {
      float s_, inv_;
      gam_row_scale(gam_ln_rowmax(v, a.d, lane), s_, inv_);
      if (live && lane == 0) a.rs[row] = inv_;
      sc2 = s_;
    }
    if (live) {
#pragma unroll
      for (int j = 0; j < GAM_LN_MAXJ; ++j) {
        const int c = (j * 64 + lane) * 4;
        if (c < a.d) gam_store4(a.out2, (size_t)row * a.d, c, v[j].x * sc2, v[j].y * sc2, v[j].z * sc2, v[j].w * sc2, a.split2);
      }
    }
  }
  if (MODE == 1) {
    float* rb = rowbuf + wave * (GAM_LN_MAXJ * 256);
#pragma unroll
    for (int j = 0; j < GAM_LN_MAXJ; ++j) {
      const int c = (j * 64 + lane) * 4;
      if (c < a.d) *reinterpret_cast<float4*>(rb + c) = v[j];
    }
    __syncthreads();
    const int half = a.dk >> 1;
    int t = a.row_t != nullptr ? a.row_t[row] : row % a.ta;
    t = t < a.rope_rows ? t : a.rope_rows - 1;
#pragma unroll
    for (int j = 0; j < GAM_LN_MAXJ; ++j) {
      const int c = (j * 64 + lane) * 4;
      if (c < a.d) {
        // c and dk/2 are multiples of 4: the 4 elements of this lane sit in the same half of one head
        float o[4];
        const int i0 = c % a.dk;
        const bool lo_half = i0 < half;
        const int po = lo_half ? half : -half;
        const float4 self4 = *reinterpret_cast<const float4*>(rb + c);
        const float4 oth4 = *reinterpret_cast<const float4*>(rb + c + po);
        const int ib = lo_half ? i0 : i0 - half;
        const float4 c4 = EARLY ? rc4[j] : *reinterpret_cast<const float4*>(a.rcos + (size_t)t * half + ib);
        const float4 s4 = EARLY ? rs4[j] : *reinterpret_cast<const float4*>(a.rsin + (size_t)t * half + ib);
        if (lo_half) {   // x*cos + (-x2)*sin
          o[0] = self4.x * c4.x - oth4.x * s4.x; o[1] = self4.y * c4.y - oth4.y * s4.y;
          o[2] = self4.z * c4.z - oth4.z * s4.z; o[3] = self4.w * c4.w - oth4.w * s4.w;
        } else {
          o[0] = self4.x * c4.x + oth4.x * s4.x; o[1] = self4.y * c4.y + oth4.y * s4.y;
          o[2] = self4.z * c4.z + oth4.z * s4.z; o[3] = self4.w * c4.w + oth4.w * s4.w;
        }
        if (live) gam_store4(a.out2, (size_t)row * a.d, c, o[0] * sc2, o[1] * sc2, o[2] * sc2, o[3] * sc2, a.split2);
      }
    }
  }
}

static inline hipError_t gam_launch_layernorm(const GamLnArgs& a, int mode, hipStream_t s) {
  if (a.rows <= 0) return hipSuccess;
  if (a.d % 4 != 0 || a.d > GAM_LN_MAXJ * 256) return hipErrorInvalidValue;
  if ((a.split1 || a.split2) && a.d % 32 != 0) return hipErrorInvalidValue;
  if (mode == 1 && (a.dk % 8 != 0 || a.d % a.dk != 0)) return hipErrorInvalidValue;   // rope: 4-element groups stay inside a half head
  if (a.part != nullptr && (a.pbias == nullptr || a.presid == nullptr || a.nsplit < 1)) return hipErrorInvalidValue;
  const int grid = gam_cdiv(a.rows, 4);
  // a few hundred rows (single clips): the launch is one latency chain per wave, not a stream -- the EARLY form (measured: one
  // 5 s clip 3.41 -> 3.34 ms; from ~4000 rows the kernel is bandwidth-bound and the lean form is the faster one)
  const bool early = a.rows <= GAM_LN_EARLY_ROWS;
#define GAM_LN_GO(MODE, PART, EARLY) hipLaunchKernelGGL((gam_layernorm_kernel<MODE, PART, EARLY>), dim3(grid), dim3(256), 0, s, a)
#define GAM_LN_MODE(PART, EARLY) \
  do { if (mode == 0) GAM_LN_GO(0, PART, EARLY); else if (mode == 1) GAM_LN_GO(1, PART, EARLY); else GAM_LN_GO(2, PART, EARLY); } while (0)
  if (a.part != nullptr) GAM_LN_MODE(true, true);      // split-K only exists on small grids
  else if (early) GAM_LN_MODE(false, true);
  else GAM_LN_MODE(false, false);
#undef GAM_LN_MODE
#undef GAM_LN_GO
  return hipGetLastError();
}
