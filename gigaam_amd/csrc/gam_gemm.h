// gam_gemm.h -- fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )        ("NT": both operands K-contiguous,
//                                                  W is a torch Linear weight as stored)
//
// Every dense contraction of the GigaAM hot path goes through this kernel
// (reference gigaam/encoder.py: Linear 12288->768 :127, FFN :419-424, q/k/v/out
// :145-148, pointwise convs :379,394; decoder.py heads), and so do the three
// convolution-shaped ones as implicit GEMMs:
//   * a_mode 1  -- the 3x3/stride-2 Conv2d #2 of the stem (encoder.py:59-69) over a
//                  zero-bordered channels-last image, K index = (kh,kw,c)
//   * a_mode 0 with lda < K -- overlapping rows: the stride-2 Conv1d stem of v3 models
//                  (K index = (k,c), lda = 2*C) and the 400-point windowed DFT of the
//                  log-mel frontend (lda = hop = 160)
//
// Arithmetic is exact fp32 (MFMA f32 == an fmaf chain): parity with the reference's
// fp32 CPU path is the first gate, and the fp32 matrix rate (157 TFLOP/s) already
// exceeds what the 2000x real-time target needs.
//
// Tiling: 128x128x32 block tile, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2
// MFMA tiles of 32x32; LDS rows padded to 36 floats (9 x 16 B, odd) so both the
// ds_write_b128 staging and the ds_read_b128 fragment reads are bank-conflict-free;
// register-staged global->LDS pipeline (next k-tile in flight in VGPRs), one 36 KB LDS
// tile set so that three workgroups share a CU.  A lane
// reads 4 consecutive k of its row with one ds_read_b128 and feeds them to 4 MFMAs
// (k-pairs {q, q+4}): the permuted k order is the same for A and W, so the dot
// product is unchanged.  blockIdx -> tile mapping is XCD-aware (all N-tiles of an
// M-tile land on the same XCD/L2, so A is fetched from HBM once).
#pragma once
#include "gam_common.h"
#include <stdlib.h>

struct GamGemmArgs {
  const float* A;
  const float* W;
  const float* bias;   // [N] or null
  const float* R;      // residual [*, ldr] or null:  C = R + alpha * act(acc + bias)
  float* C;
  int M, N, K;         // K % 32 == 0
  long lda, ldc, ldr;
  float alpha;
  // A addressing
  int a_mode;          // 0: row m at A + m*lda ; 1: conv2d-stem gather (see above)
  int conv_fp;         // a_mode 1: padded feature count of the image (F1 + 1)
  int conv_c;          // a_mode 1: channels of the image
  int conv_f2;         // a_mode 1: output features per frame (m = (b*Ta + t2)*f2 + f)
  // row mask: b = m / rpb, t = (m % rpb) / fdiv; rows with t >= lens[b] are written as 0
  const int* lens;
  int rpb, fdiv;
  // row remap: out_row = (m / rpb) * out_rpb + (m % rpb) + out_shift; rows with
  // (m % rpb) >= rows_valid are skipped
  int remap, out_rpb, out_shift, rows_valid;
  // packed-row batches (gam_pack.h): the stem still runs on the padded layout, but a row tile that lies wholly inside ONE utterance's
  // padding frames (t >= lens[b] for all its rows) produces nothing the gather behind the stem reads -- gam_gemm_sp_kernel returns at once
  int skip_pad;
  // split-fp16 operand planes of W (gam_gemm16.h): W * 2^wshift = Whi + Wlo (+ ~2^-22 |W|)
  const _Float16* Whi;
  const _Float16* Wlo;
  // both operands in the sp32 layout of gam_gemm_sp.h (row pitch = lda / K elements, 4 B each)
  const _Float16* Asp;
  const _Float16* Wsp;
  // gam_gemm_sp.h only: output columns n >= n_switch (> 0, a multiple of the tile width) contract Asp2 instead of Asp -- one
  // launch for two projections of two operands that share rows, pitch and row scale (q|k of the rotated copy, v of the plain one)
  const _Float16* Asp2;
  int n_switch;
  int c_split;          // 1: write C in the sp32 layout (row pitch ldc elements) instead of fp32; 2: as plain fp16 rows (gam_common.h gam_store4)
  // split-K (small grids): grid.y = splitk slices of K; this struct's K is the slice length, ldw the
  // full row pitch of W; slice s reads columns [s*K, (s+1)*K) and writes its raw partial sums to
  // partial[s][M][N]; gam_splitk_reduce_kernel applies the epilogue.  0 / 1 = off.
  int splitk;
  long ldw;             // row pitch of W / Whi / Wlo in elements (0 = K)
  float* partial;
  int sp_mt, sp_nw;     // LDS-DMA GEMM: the plan's tile shape (gam_gemm_sp_plan); 0 = let the launcher plan (no split-K)
  int sp_ns;            // LDS stages of the plan (2 or 3)
  int a_fmt;            // format of the Asp operand as its producer wrote it: 1 = sp32 (hi, lo) lines, 2 = plain fp16 rows
  int h16;              // GAM_GEMM_F16 (opt-in speed mode): plain-fp16 operands, one MFMA per product; K / lda / conv_c are HALVED by the caller
  int ntiles;           // set by the launcher
  int dbg;              // experiment switches (GAM_SP_DBG), 0 in production
  float wscale_inv;     // 2^-wshift, applied to the accumulator in the epilogue
  // split-fp16 modes: per-row pre-scale of the A operand (gam_common.h gam_row_scale).  The producing kernel stored row m
  // multiplied by 2^e_m (sp32 or fp32 alike); a_rs[m] = 2^-e_m multiplies row m's accumulators in the epilogue.
  // null = none (always null on the exact-fp32 path).
  const float* a_rs;
  // Range guard: C feeds a split-fp16 GEMM as an UNSCALED operand (FFN hidden, stem activations).  With c_guard set, a
  // value beyond fp16's range written to C sets *range_flag (gam_common.h gam_range_note); may be null.
  int* range_flag;
  int c_guard;
  long long* tlog;      // instrumented builds (GAM_SP_INSTRUMENT, GAM_SP_DBG & 16): per-workgroup timeline, 8 words per tile
  int prio;             // LDS-DMA GEMM: s_setprio 1 for the later-dispatched half of the waves (GAM_SP_PRIO, A/B switch)
  int tile_order;       // LDS-DMA GEMM: 0 = n innermost (the default), G > 0 = groups of G column tiles outermost (GAM_SP_ORDER, r05 experiment)
};

#define GAM_GEMM_BM 128
#define GAM_GEMM_BN 128
#define GAM_GEMM_BK 32
#define GAM_GEMM_LD 36
#define GAM_GEMM_SMEM(NBUF) ((NBUF) * (GAM_GEMM_BM + GAM_GEMM_BN) * GAM_GEMM_LD * 4)

// ---- shared epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)
template <int ACT>
__device__ __forceinline__ void gam_gemm_epilogue(const GamGemmArgs& g, const f32x16& acc00, const f32x16& acc01,
                                                  const f32x16& acc10, const f32x16& acc11, int m0, int n0, int wm,
                                                  int wn, int lane, float accscale, const float* rs_lds = nullptr) {
  // rs_lds: the tile's per-row A-operand factors (a_rs[m0 + i]) staged in LDS by the caller, or null (none).  (They were
  // first loaded into a 16-register array per 32-row slab here: +32 VGPRs, and the pipelined 128x128 kernel dropped
  // from 3 to 2 waves per SIMD -- a single 5 s clip went from 3.6 to 4.6 ms.)
  const int lcol = lane & 31;
  const int lrow4 = 4 * (lane >> 5);
  if (g.partial != nullptr) {   // split-K slice: raw partial sums, the reduce kernel finishes
    float* P = g.partial + (size_t)blockIdx.y * (size_t)g.M * (size_t)g.N;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + lrow4;
        if (row >= g.M) continue;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          const int col = n0 + wn * 64 + tn * 32 + lcol;
          if (col < g.N)
            P[(size_t)row * g.N + col] = (tm == 0 ? (tn == 0 ? acc00[r] : acc01[r]) : (tn == 0 ? acc10[r] : acc11[r])) *
                                         (accscale * (rs_lds != nullptr ? rs_lds[row - m0] : 1.0f));
        }
      }
    return;
  }
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + lrow4;
      if (row >= g.M) continue;
      bool masked = false;
      long orow = row;
      if (g.lens != nullptr || g.remap) {
        const int bb = row / g.rpb, tt = row - bb * g.rpb;
        if (g.lens != nullptr) masked = (tt / g.fdiv) >= g.lens[bb];
        if (g.remap) {
          if (tt >= g.rows_valid) continue;
          orow = (long)bb * g.out_rpb + tt + g.out_shift;
        }
      }
      const float rowscale = rs_lds != nullptr ? accscale * rs_lds[row - m0] : accscale;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int col = n0 + wn * 64 + tn * 32 + lcol;
        if (col >= g.N) continue;
        float v = (tm == 0 ? (tn == 0 ? acc00[r] : acc01[r]) : (tn == 0 ? acc10[r] : acc11[r])) * rowscale;
        if (g.bias != nullptr) v += g.bias[col];
        if (ACT == GAM_ACT_SILU) v = gam_silu(v);
        if (ACT == GAM_ACT_RELU) v = fmaxf(v, 0.0f);
        if (masked) v = 0.0f;
        v *= g.alpha;
        if (g.R != nullptr) v += g.R[orow * g.ldr + col];
        if (g.c_guard) gam_range_note(g.range_flag, v, 0.f, 0.f, 0.f);
        g.C[orow * g.ldc + col] = v;
      }
    }
  }
}

template <int ACT, int NBUF>
__global__ __launch_bounds__(256, NBUF == 1 ? 3 : 2) void gam_gemm_f32_kernel(GamGemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float gam_smem[];
  if (g.splitk > 1) { const size_t ko = (size_t)blockIdx.y * (size_t)g.K; g.A += ko; g.W += ko; }
  constexpr int BM = GAM_GEMM_BM, BN = GAM_GEMM_BN, BK = GAM_GEMM_BK, LD = GAM_GEMM_LD;
  float* As = gam_smem;
  float* Bs = gam_smem + NBUF * BM * LD;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- XCD-aware block -> tile map (bijective for any grid size) ----
  const int nbn = (g.N + BN - 1) / BN;
  const int total = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = total >> 3, r8 = total & 7;
  const int xcd = bid & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int m0 = (lid / nbn) * BM;
  const int n0 = (lid % nbn) * BN;

  // ---- per-thread global load coordinates: 4 rows of A, 4 rows of W, one float4 each ----
  // (named scalars, not arrays: arrays captured by the staging code end up in scratch)
  const int lrow = tid >> 3;
  const int lc4 = (tid & 7) * 4;
  auto a_row_off = [&](int i) -> size_t {
    int m = m0 + lrow + 32 * i;
    m = m < g.M ? m : g.M - 1;
    if (g.a_mode == 0) return (size_t)m * (size_t)g.lda;
    const int fr = m / g.conv_f2, f = m - fr * g.conv_f2;
    return ((size_t)fr * 2 * g.conv_fp + 2 * f) * (size_t)g.conv_c;
  };
  auto w_row_off = [&](int i) -> size_t {
    int n = n0 + lrow + 32 * i;
    n = n < g.N ? n : g.N - 1;
    return (size_t)n * (size_t)g.ldw;
  };
  const float* pa0 = g.A + a_row_off(0) + lc4;
  const float* pa1 = g.A + a_row_off(1) + lc4;
  const float* pa2 = g.A + a_row_off(2) + lc4;
  const float* pa3 = g.A + a_row_off(3) + lc4;
  const float* pw0 = g.W + w_row_off(0) + lc4;
  const float* pw1 = g.W + w_row_off(1) + lc4;
  const float* pw2 = g.W + w_row_off(2) + lc4;
  const float* pw3 = g.W + w_row_off(3) + lc4;

  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }

  const int nk = g.K / BK;
  float4 va0, va1, va2, va3, vb0, vb1, vb2, vb3;

#define GAM_GLOAD(KT)                                                           \
  {                                                                             \
    const int k0_ = (KT) * BK;                                                  \
    size_t ka_ = (size_t)k0_;                                                   \
    if (g.a_mode != 0) {                                                        \
      const int tap_ = k0_ / g.conv_c, c0_ = k0_ - tap_ * g.conv_c;             \
      const int kh_ = tap_ / 3, kw_ = tap_ - kh_ * 3;                           \
      ka_ = ((size_t)kh_ * g.conv_fp + kw_) * (size_t)g.conv_c + c0_;           \
    }                                                                           \
    va0 = *reinterpret_cast<const float4*>(pa0 + ka_);                          \
    va1 = *reinterpret_cast<const float4*>(pa1 + ka_);                          \
    va2 = *reinterpret_cast<const float4*>(pa2 + ka_);                          \
    va3 = *reinterpret_cast<const float4*>(pa3 + ka_);                          \
    vb0 = *reinterpret_cast<const float4*>(pw0 + k0_);                          \
    vb1 = *reinterpret_cast<const float4*>(pw1 + k0_);                          \
    vb2 = *reinterpret_cast<const float4*>(pw2 + k0_);                          \
    vb3 = *reinterpret_cast<const float4*>(pw3 + k0_);                          \
  }
#define GAM_LSTORE(BUF)                                                         \
  {                                                                             \
    float* a_ = As + (BUF) * BM * LD + lrow * LD + lc4;                         \
    float* b_ = Bs + (BUF) * BN * LD + lrow * LD + lc4;                         \
    *reinterpret_cast<float4*>(a_) = va0;                                       \
    *reinterpret_cast<float4*>(a_ + 32 * LD) = va1;                             \
    *reinterpret_cast<float4*>(a_ + 64 * LD) = va2;                             \
    *reinterpret_cast<float4*>(a_ + 96 * LD) = va3;                             \
    *reinterpret_cast<float4*>(b_) = vb0;                                       \
    *reinterpret_cast<float4*>(b_ + 32 * LD) = vb1;                             \
    *reinterpret_cast<float4*>(b_ + 64 * LD) = vb2;                             \
    *reinterpret_cast<float4*>(b_ + 96 * LD) = vb3;                             \
  }
#define GAM_MFMA(AV, BV, ACC) ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(AV, BV, ACC, 0, 0, 0)

  // One LDS tile set (36 KB -> 3 workgroups per CU, i.e. 768 resident tiles: the
  // 126 x 6 = 756 tiles of an N = 768 GEMM at the bench shape run in ONE round), the
  // next k-tile travels through registers while the current one is multiplied:
  //   barrier | regs -> LDS | barrier | issue global loads of tile kt+1 | 64 MFMAs
  // The two barriers per k-tile are covered by the other resident workgroups' MFMAs.
  const int frag = (lane & 31) * LD + (lane >> 5) * 4;
#define GAM_COMPUTE(AB, BB)                                                                     \
  _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                               \
    const float4 a0 = *reinterpret_cast<const float4*>((AB) + s * 8);                           \
    const float4 a1 = *reinterpret_cast<const float4*>((AB) + 32 * LD + s * 8);                 \
    const float4 b0 = *reinterpret_cast<const float4*>((BB) + s * 8);                           \
    const float4 b1 = *reinterpret_cast<const float4*>((BB) + 32 * LD + s * 8);                 \
    /* four independent accumulators: dependent MFMAs are 4 issues apart */                     \
    GAM_MFMA(a0.x, b0.x, acc00); GAM_MFMA(a0.x, b1.x, acc01); GAM_MFMA(a1.x, b0.x, acc10); GAM_MFMA(a1.x, b1.x, acc11); \
    GAM_MFMA(a0.y, b0.y, acc00); GAM_MFMA(a0.y, b1.y, acc01); GAM_MFMA(a1.y, b0.y, acc10); GAM_MFMA(a1.y, b1.y, acc11); \
    GAM_MFMA(a0.z, b0.z, acc00); GAM_MFMA(a0.z, b1.z, acc01); GAM_MFMA(a1.z, b0.z, acc10); GAM_MFMA(a1.z, b1.z, acc11); \
    GAM_MFMA(a0.w, b0.w, acc00); GAM_MFMA(a0.w, b1.w, acc01); GAM_MFMA(a1.w, b0.w, acc10); GAM_MFMA(a1.w, b1.w, acc11); \
  }
  if constexpr (NBUF == 1) {
    // One LDS tile set (36 KB -> 3 workgroups per CU, i.e. 768 resident tiles: the
    // 126 x 6 = 756 tiles of an N = 768 GEMM at the bench shape run in ONE round); the
    // next k-tile travels through registers while the current one is multiplied:
    //   barrier | regs -> LDS | barrier | issue global loads of tile kt+1 | 64 MFMAs
    // The two barriers per k-tile are covered by the other resident workgroups' MFMAs.
    GAM_GLOAD(0);
    const float* Ab = As + wm * 64 * LD + frag;
    const float* Bb = Bs + wn * 64 * LD + frag;
    for (int kt = 0; kt < nk; ++kt) {
      __syncthreads();
      GAM_LSTORE(0);
      __syncthreads();
      if (kt + 1 < nk) GAM_GLOAD(kt + 1);
      GAM_COMPUTE(Ab, Bb);
    }
  } else {
    // Two LDS tile sets (72 KB -> 2 workgroups per CU), one barrier per k-tile.
    GAM_GLOAD(0);
    GAM_LSTORE(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      const bool more = kt + 1 < nk;
      if (more) GAM_GLOAD(kt + 1);
      const float* Ab = As + buf * BM * LD + wm * 64 * LD + frag;
      const float* Bb = Bs + buf * BN * LD + wn * 64 * LD + frag;
      GAM_COMPUTE(Ab, Bb);
      if (more) GAM_LSTORE(buf ^ 1);
      __syncthreads();
    }
  }
#undef GAM_COMPUTE
#undef GAM_GLOAD
#undef GAM_LSTORE
#undef GAM_MFMA

  gam_gemm_epilogue<ACT>(g, acc00, acc01, acc10, acc11, m0, n0, wm, wn, lane, 1.0f);
}

template <int ACT, int NBUF>
static inline void gam_launch_gemm_t(const GamGemmArgs& a, int grid, hipStream_t stream) {
  static std::atomic<unsigned long long> attr_devs{0};
  if (gam_set_max_lds(reinterpret_cast<const void*>(gam_gemm_f32_kernel<ACT, NBUF>), GAM_GEMM_SMEM(NBUF), attr_devs) != hipSuccess) return;
  hipLaunchKernelGGL((gam_gemm_f32_kernel<ACT, NBUF>), dim3(grid, a.splitk > 1 ? a.splitk : 1), dim3(256), GAM_GEMM_SMEM(NBUF), stream, a);
}

// GAM_GEMM_NBUF=2 selects the double-buffered variant (A/B measurements only)
static inline int gam_gemm_nbuf() {
  static int v = 0;
  if (v == 0) {
    const char* e = getenv("GAM_GEMM_NBUF");
    v = (e && e[0] == '2') ? 2 : 1;
  }
  return v;
}

static inline hipError_t gam_launch_gemm(const GamGemmArgs& a_in, int act, hipStream_t stream) {
  GamGemmArgs a = a_in;
  if (a.ldw == 0) a.ldw = a.K;
  if (a.M <= 0 || a.N <= 0) return hipSuccess;
  if (a.K % GAM_GEMM_BK != 0 || a.K <= 0) return hipErrorInvalidValue;
  const int grid = gam_cdiv(a.M, GAM_GEMM_BM) * gam_cdiv(a.N, GAM_GEMM_BN);
  const bool one = gam_gemm_nbuf() == 1;
  switch (act) {
    case GAM_ACT_SILU:
      if (one) gam_launch_gemm_t<GAM_ACT_SILU, 1>(a, grid, stream); else gam_launch_gemm_t<GAM_ACT_SILU, 2>(a, grid, stream);
      break;
    case GAM_ACT_RELU:
      if (one) gam_launch_gemm_t<GAM_ACT_RELU, 1>(a, grid, stream); else gam_launch_gemm_t<GAM_ACT_RELU, 2>(a, grid, stream);
      break;
    default:
      if (one) gam_launch_gemm_t<GAM_ACT_NONE, 1>(a, grid, stream); else gam_launch_gemm_t<GAM_ACT_NONE, 2>(a, grid, stream);
      break;
  }
  return hipGetLastError();
}

// ---- split-K second pass: C = epilogue(sum_s partial[s]) with the same epilogue semantics as above.
//      One thread per 4 consecutive columns (N % 4 == 0) or per column.
template <int ACT, int VEC, int SMAX = 16>   // SMAX: largest S with all slices in flight (4: the lean instantiation for S <= 4)
__global__ __launch_bounds__(256) void gam_splitk_reduce_kernel(GamGemmArgs g, int S) {
  const int cpr = (g.N + VEC - 1) / VEC;                       // column groups per row
  const size_t total = (size_t)g.M * cpr;
  const size_t slice = (size_t)g.M * (size_t)g.N;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / cpr), col = (int)(i % cpr) * VEC;
    float v[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = 0.f;
    const float* p = g.partial + (size_t)row * g.N + col;
    if (VEC == 4) {   // all S slices in flight, summed in slice order (gam_common.h)
      const f32x4 t = gam_sum_slices<SMAX>(p, slice, S);
      v[0] = t.x; v[1 % VEC] = t.y; v[2 % VEC] = t.z; v[3 % VEC] = t.w;
    } else {
      for (int s = 0; s < S; ++s) v[0] += p[s * slice];
    }
    bool masked = false;
    long orow = row;
    if (g.lens != nullptr || g.remap) {
      const int bb = row / g.rpb, tt = row - bb * g.rpb;
      if (g.lens != nullptr) masked = (tt / g.fdiv) >= g.lens[bb];
      if (g.remap) {
        if (tt >= g.rows_valid) continue;
        orow = (long)bb * g.out_rpb + tt + g.out_shift;
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float x = v[e];
      if (g.bias != nullptr) x += g.bias[col + e];
      if (ACT == GAM_ACT_SILU) x = gam_silu(x);
      if (ACT == GAM_ACT_RELU) x = fmaxf(x, 0.0f);
      if (masked) x = 0.0f;
      x *= g.alpha;
      if (g.R != nullptr) x += g.R[orow * g.ldr + col + e];
      v[e] = x;
    }
    if (VEC == 4) {
      if (g.c_guard) gam_range_note(g.range_flag, v[0], v[1 % VEC], v[2 % VEC], v[3 % VEC]);
      if (g.c_split) gam_store4(g.C, (size_t)orow * g.ldc, col, v[0], v[1 % VEC], v[2 % VEC], v[3 % VEC], g.c_split);
      else *reinterpret_cast<f32x4*>(g.C + orow * g.ldc + col) = (f32x4){v[0], v[1 % VEC], v[2 % VEC], v[3 % VEC]};
    } else {
      if (g.c_guard) gam_range_note(g.range_flag, v[0], 0.f, 0.f, 0.f);
      gam_store1(g.C, (size_t)orow * g.ldc, col, v[0], g.c_split);
    }
  }
}

static inline hipError_t gam_launch_splitk_reduce(const GamGemmArgs& a, int act, int S, hipStream_t stream) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool vec = a.N % 4 == 0 && a.ldc % 4 == 0 && al16(a.C) && al16(a.partial) &&
                   (a.R == nullptr || (a.ldr % 4 == 0 && al16(a.R)));
  const size_t total = (size_t)a.M * (vec ? a.N / 4 : a.N);
  const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
#define GAM_RED(ACTV)                                                                                      \
  if (vec && S <= 4) hipLaunchKernelGGL((gam_splitk_reduce_kernel<ACTV, 4, 4>), dim3(grid), dim3(256), 0, stream, a, S); \
  else if (vec) hipLaunchKernelGGL((gam_splitk_reduce_kernel<ACTV, 4, 16>), dim3(grid), dim3(256), 0, stream, a, S); \
  else hipLaunchKernelGGL((gam_splitk_reduce_kernel<ACTV, 1>), dim3(grid), dim3(256), 0, stream, a, S);
  switch (act) {
    case GAM_ACT_SILU: GAM_RED(GAM_ACT_SILU); break;
    case GAM_ACT_RELU: GAM_RED(GAM_ACT_RELU); break;
    default: GAM_RED(GAM_ACT_NONE); break;
  }
#undef GAM_RED
  return hipGetLastError();
}
