// gam_gemm16.h -- fp32-accurate GEMM on the fp16 matrix cores: three-term split
// ("f16x3") on v_mfma_f32_32x32x16_f16, 16x the per-instruction rate of the fp32 MFMA.
//
//   a = a_hi + a_lo,  w * 2^s = w_hi + w_lo      (hi = fp16(x), lo = fp16(x - hi))
//   a.w ~= (a_hi.w_hi + a_hi.w_lo + a_lo.w_hi) * 2^-s          accumulated in fp32
//
// fp16 carries 11 significant bits, so hi+lo keeps 22 of fp32's 24 and the dropped
// a_lo.w_lo term is ~2^-22 relative: measured on the full 16-layer encoder the split
// path is as close to an fp64 evaluation as plain fp32 is (3.8e-6 vs 3.1e-6 max abs,
// DESIGN.md §numerics).  The power-of-two pre-scale of each weight matrix (exact,
// undone in the epilogue) keeps w_lo clear of the fp16 subnormal range; activations are
// O(1) after LayerNorm / SiLU and need none.  W planes are built once at gam_finalize;
// A is split on the fly while it is staged global -> registers -> LDS.
//
// Same 128x128 block tile / 2x2 waves / XCD-aware tile map / fused epilogue as
// gam_gemm.h.  LDS holds four fp16 planes (A_hi, A_lo, W_hi, W_lo) with rows padded
// to an odd number of 16-byte slots, so a lane's 8-element MFMA fragment is one
// conflict-free ds_read_b128.
#pragma once
#include "gam_gemm.h"
#include <type_traits>


template <int BK, int BM = 128>
struct GamGemm16Cfg {
  static constexpr int NT = 2 * BM;                      // threads: BM/64 x 2 waves of 64x64
  static constexpr int LD = BK + 8;                      // halfs per LDS row (80 B / 144 B: odd # of 16 B slots)
  static constexpr int APLANE = BM * LD;                 // halfs per A plane
  static constexpr int WPLANE = 128 * LD;                // halfs per W plane
  static constexpr int STAGE = 2 * (APLANE + WPLANE);    // halfs per LDS stage (4 planes)
  static constexpr int SMEM = STAGE * 2;                 // bytes per stage
  static constexpr int A_F4 = BM * BK / 4 / NT;          // float4 loads of A per thread per k-tile
  static constexpr int W_CH = 128 * BK / 8 / NT;         // 16-byte chunks per thread per W plane per k-tile
};


template <int ACT, int BM, bool PIPE>
__global__ __launch_bounds__(2 * BM, PIPE ? 2 : (BM == 256 ? 4 : 3)) void gam_gemm_f16x3_kernel(GamGemmArgs g) {
  constexpr int BK = 32;
  extern __shared__ __attribute__((aligned(16))) _Float16 gam_smem16[];
  if (g.splitk > 1) {
    const size_t ko = (size_t)blockIdx.y * (size_t)g.K;
    g.A += ko; g.Whi += ko; g.Wlo += ko;
  }
  using Cfg = GamGemm16Cfg<BK, BM>;
  constexpr int BN = 128, LD = Cfg::LD, NT = Cfg::NT;
  _Float16* Ahi = gam_smem16;
  _Float16* Alo = gam_smem16 + Cfg::APLANE;
  _Float16* Whi = gam_smem16 + 2 * Cfg::APLANE;
  _Float16* Wlo = gam_smem16 + 2 * Cfg::APLANE + Cfg::WPLANE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int nbn = (g.N + BN - 1) / BN;
  const int total = gridDim.x;
  const int q8 = total >> 3, r8 = total & 7;
  const int bid = blockIdx.x;
  const int xcd = bid & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int m0 = (lid / nbn) * BM;
  const int n0 = (lid % nbn) * BN;

  // ---- A staging: float4 index f = tid + 256*i over [128 rows][BK/4];  W staging: 16 B chunk
  //      c = tid + 256*i over [128 rows][BK/8] per plane
  constexpr int AF = Cfg::A_F4, WC = Cfg::W_CH, F4R = BK / 4, CHR = BK / 8;
  constexpr int AEL = 4;   // elements per A staging item (fp32 A, split while it is staged)
  size_t a_off[AF];
  int a_lds[AF];
#pragma unroll
  for (int i = 0; i < AF; ++i) {
    const int f = tid + NT * i;
    const int row = f / F4R, c4 = (f % F4R) * AEL;
    int m = m0 + row;
    m = m < g.M ? m : g.M - 1;
    size_t off;
    if (g.a_mode == 0) off = (size_t)m * (size_t)g.lda;
    else {
      const int fr = m / g.conv_f2, ff = m - fr * g.conv_f2;
      off = ((size_t)fr * 2 * g.conv_fp + 2 * ff) * (size_t)g.conv_c;
    }
    a_off[i] = off + c4;
    a_lds[i] = row * LD + c4;
  }
  size_t w_off[WC];
  int w_lds[WC];
#pragma unroll
  for (int i = 0; i < WC; ++i) {
    const int c = tid + NT * i;
    const int row = c / CHR, part = (c % CHR) * 8;
    int n = n0 + row;
    n = n < g.N ? n : g.N - 1;
    w_off[i] = (size_t)n * (size_t)g.ldw + part;
    w_lds[i] = row * LD + part;
  }

  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }

  const int nk = g.K / BK;
  f32x4 va[AF];
  gam_u32x4 vh[WC], vl[WC];

  auto gload = [&](int kt) {
    const int k0 = kt * BK;
    size_t ka = (size_t)k0;
    if (g.a_mode != 0) {
      const int tap = k0 / g.conv_c, c0 = k0 - tap * g.conv_c;
      const int kh = tap / 3, kw = tap - kh * 3;
      ka = ((size_t)kh * g.conv_fp + kw) * (size_t)g.conv_c + c0;
    }
#pragma unroll
    for (int i = 0; i < AF; ++i) va[i] = *reinterpret_cast<const f32x4*>(g.A + a_off[i] + ka);
#pragma unroll
    for (int i = 0; i < WC; ++i) {
      vh[i] = *reinterpret_cast<const gam_u32x4*>(g.Whi + w_off[i] + k0);
      vl[i] = *reinterpret_cast<const gam_u32x4*>(g.Wlo + w_off[i] + k0);
    }
  };
  auto lstore = [&](int buf) {
    _Float16* ah = Ahi + buf * Cfg::STAGE;
    _Float16* al = Alo + buf * Cfg::STAGE;
    _Float16* wh = Whi + buf * Cfg::STAGE;
    _Float16* wl = Wlo + buf * Cfg::STAGE;
#pragma unroll
    for (int i = 0; i < AF; ++i) {
      gam_half4 hi, lo;
      gam_split4(va[i], hi, lo);
      *reinterpret_cast<gam_half4*>(ah + a_lds[i]) = hi;
      *reinterpret_cast<gam_half4*>(al + a_lds[i]) = lo;
    }
#pragma unroll
    for (int i = 0; i < WC; ++i) {
      *reinterpret_cast<gam_u32x4*>(wh + w_lds[i]) = vh[i];
      *reinterpret_cast<gam_u32x4*>(wl + w_lds[i]) = vl[i];
    }
  };

#define GAM_MF16(AV, BV, ACC) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AV, BV, ACC, 0, 0, 0)
#define GAM_K16(BUF, KS)                                                                                        \
  {                                                                                                             \
    const int o_ = (BUF) * Cfg::STAGE + (KS) * 16;                                                              \
    const gam_half8 ah0 = *reinterpret_cast<const gam_half8*>(Ahi + o_ + fa);                                  \
    const gam_half8 ah1 = *reinterpret_cast<const gam_half8*>(Ahi + o_ + fa + 32 * LD);                        \
    const gam_half8 bh0 = *reinterpret_cast<const gam_half8*>(Whi + o_ + fb);                                  \
    const gam_half8 bh1 = *reinterpret_cast<const gam_half8*>(Whi + o_ + fb + 32 * LD);                        \
    GAM_MF16(ah0, bh0, acc00); GAM_MF16(ah0, bh1, acc01); GAM_MF16(ah1, bh0, acc10); GAM_MF16(ah1, bh1, acc11); \
    const gam_half8 bl0 = *reinterpret_cast<const gam_half8*>(Wlo + o_ + fb);                                  \
    const gam_half8 bl1 = *reinterpret_cast<const gam_half8*>(Wlo + o_ + fb + 32 * LD);                        \
    GAM_MF16(ah0, bl0, acc00); GAM_MF16(ah0, bl1, acc01); GAM_MF16(ah1, bl0, acc10); GAM_MF16(ah1, bl1, acc11); \
    const gam_half8 al0 = *reinterpret_cast<const gam_half8*>(Alo + o_ + fa);                                  \
    const gam_half8 al1 = *reinterpret_cast<const gam_half8*>(Alo + o_ + fa + 32 * LD);                        \
    GAM_MF16(al0, bh0, acc00); GAM_MF16(al0, bh1, acc01); GAM_MF16(al1, bh0, acc10); GAM_MF16(al1, bh1, acc11); \
  }
  const int frag = (lane & 31) * LD + (lane >> 5) * 8;
  const int fa = wm * 64 * LD + frag, fb = wn * 64 * LD + frag;
  if constexpr (PIPE) {
    // Two LDS stages, one barrier per k-tile.  While tile kt is multiplied, the registers that
    // hold tile kt+1 are split / written to the other stage and immediately re-loaded with
    // tile kt+2, one staging item after every third MFMA, so the VALU, ds_write and
    // global_load issue slots fall into the shadow of MFMAs in flight.  (With separate load /
    // store / multiply phases the co-resident workgroups run phase-aligned and the three costs
    // add up: measured 614 + 369 + 163 us at K = 12288.)
    auto item = [&](auto st, auto ld, int j, int sbuf, size_t ka, int k0) {
      // staging item j: 0..AF-1 = A float4 j ; AF..AF+2*WC-1 = W chunk (hi/lo interleaved)
      if (j < AF) {
        if constexpr (decltype(st)::value) {
          gam_half4 hi, lo;
          gam_split4(va[j], hi, lo);
          *reinterpret_cast<gam_half4*>(Ahi + sbuf * Cfg::STAGE + a_lds[j]) = hi;
          *reinterpret_cast<gam_half4*>(Alo + sbuf * Cfg::STAGE + a_lds[j]) = lo;
        }
        if constexpr (decltype(ld)::value) va[j] = *reinterpret_cast<const f32x4*>(g.A + a_off[j] + ka);
      } else {
        const int c = (j - AF) >> 1;
        if ((j - AF) & 1) {
          if constexpr (decltype(st)::value) *reinterpret_cast<gam_u32x4*>(Wlo + sbuf * Cfg::STAGE + w_lds[c]) = vl[c];
          if constexpr (decltype(ld)::value) vl[c] = *reinterpret_cast<const gam_u32x4*>(g.Wlo + w_off[c] + k0);
        } else {
          if constexpr (decltype(st)::value) *reinterpret_cast<gam_u32x4*>(Whi + sbuf * Cfg::STAGE + w_lds[c]) = vh[c];
          if constexpr (decltype(ld)::value) vh[c] = *reinterpret_cast<const gam_u32x4*>(g.Whi + w_off[c] + k0);
        }
      }
    };
    auto body = [&](auto st, auto ld, int kt) {
      const int cur = kt & 1, sbuf = cur ^ 1;
      const int k0 = (kt + 2) * BK;
      size_t ka = (size_t)k0;
      if (g.a_mode != 0) {
        const int tap = k0 / g.conv_c, c0 = k0 - tap * g.conv_c;
        const int kh = tap / 3, kw = tap - kh * 3;
        ka = ((size_t)kh * g.conv_fp + kw) * (size_t)g.conv_c + c0;
      }
      constexpr int NI = AF + 2 * WC;   // 8 staging items over 24 MFMAs
      static_assert(NI == 8, "item schedule below assumes 8 staging items");
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int o_ = cur * Cfg::STAGE + ks * 16;
        const gam_half8 ah0 = *reinterpret_cast<const gam_half8*>(Ahi + o_ + fa);
        const gam_half8 ah1 = *reinterpret_cast<const gam_half8*>(Ahi + o_ + fa + 32 * LD);
        const gam_half8 bh0 = *reinterpret_cast<const gam_half8*>(Whi + o_ + fb);
        const gam_half8 bh1 = *reinterpret_cast<const gam_half8*>(Whi + o_ + fb + 32 * LD);
        const gam_half8 bl0 = *reinterpret_cast<const gam_half8*>(Wlo + o_ + fb);
        const gam_half8 bl1 = *reinterpret_cast<const gam_half8*>(Wlo + o_ + fb + 32 * LD);
        const gam_half8 al0 = *reinterpret_cast<const gam_half8*>(Alo + o_ + fa);
        const gam_half8 al1 = *reinterpret_cast<const gam_half8*>(Alo + o_ + fa + 32 * LD);
        GAM_MF16(ah0, bh0, acc00); GAM_MF16(ah0, bh1, acc01); GAM_MF16(ah1, bh0, acc10);
        item(st, ld, ks * 4 + 0, sbuf, ka, k0);
        GAM_MF16(ah1, bh1, acc11); GAM_MF16(ah0, bl0, acc00); GAM_MF16(ah0, bl1, acc01);
        item(st, ld, ks * 4 + 1, sbuf, ka, k0);
        GAM_MF16(ah1, bl0, acc10); GAM_MF16(ah1, bl1, acc11); GAM_MF16(al0, bh0, acc00);
        item(st, ld, ks * 4 + 2, sbuf, ka, k0);
        GAM_MF16(al0, bh1, acc01); GAM_MF16(al1, bh0, acc10); GAM_MF16(al1, bh1, acc11);
        item(st, ld, ks * 4 + 3, sbuf, ka, k0);
      }
      __syncthreads();
    };
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;
    gload(0);
    lstore(0);
    if (nk > 1) gload(1);
    __syncthreads();
    int kt = 0;
    for (; kt + 2 < nk; ++kt) body(T_{}, T_{}, kt);
    if (kt + 1 < nk) { body(T_{}, F_{}, kt); ++kt; }
    body(F_{}, F_{}, kt);
  } else {
    gload(0);
    for (int kt = 0; kt < nk; ++kt) {
      __syncthreads();
      lstore(0);
      __syncthreads();
      if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) GAM_K16(0, ks);
    }
  }
#undef GAM_K16
#undef GAM_MF16
  // the tile's row factors go through LDS (the stages are dead): one global load per row, no register array
  const float* rs_lds = nullptr;
  if (g.a_rs != nullptr) {
    float* rl = reinterpret_cast<float*>(gam_smem16);
    __syncthreads();                       // every wave is done reading the last k-tile
    if (tid < BM) { const int m = m0 + tid; rl[tid] = g.a_rs[m < g.M ? m : g.M - 1]; }
    __syncthreads();
    rs_lds = rl;
  }
  gam_gemm_epilogue<ACT>(g, acc00, acc01, acc10, acc11, m0, n0, wm, wn, lane, g.wscale_inv, rs_lds);
}

template <int ACT, int BM = 128, bool PIPE = false>
static inline void gam_launch_gemm16_t(const GamGemmArgs& a, int grid, hipStream_t stream) {
  static std::atomic<unsigned long long> attr_devs{0};
  constexpr int smem = GamGemm16Cfg<32, BM>::SMEM * (PIPE ? 2 : 1);
  auto kern = gam_gemm_f16x3_kernel<ACT, BM, PIPE>;
  if (gam_set_max_lds(reinterpret_cast<const void*>(kern), smem, attr_devs) != hipSuccess) return;   // the error stays latched for hipGetLastError()
  hipLaunchKernelGGL(kern, dim3(grid, a.splitk > 1 ? a.splitk : 1), dim3(2 * BM), smem, stream, a);
}

// fp32 -> (hi, lo) fp16 planes over a flat range (elementwise, HBM-bound: 4 B in, 4 B out)
__global__ __launch_bounds__(256) void gam_split_kernel(const float* __restrict__ x, _Float16* __restrict__ hi,
                                                        _Float16* __restrict__ lo, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    gam_half4 h, l;
    gam_split4(v, h, l);
    reinterpret_cast<gam_half4*>(hi)[i] = h;
    reinterpret_cast<gam_half4*>(lo)[i] = l;
  }
}

static inline hipError_t gam_launch_gemm16(const GamGemmArgs& a_in, int act, hipStream_t stream) {
  GamGemmArgs a = a_in;
  if (a.ldw == 0) a.ldw = a.K;
  if (a.M <= 0 || a.N <= 0) return hipSuccess;
  if (a.K <= 0 || a.Whi == nullptr || a.Wlo == nullptr) return hipErrorInvalidValue;
  // (r05: the 256x128 8-wave instantiations of this family are gone -- they spilled 4 VGPRs and were reachable only with
  //  the LDS-DMA family switched off: every shape they took, K % 32 == 0 at >= 1000 tiles, is gam_gemm_sp_kernel's.)
  a.ntiles = gam_cdiv(a.M, 128) * gam_cdiv(a.N, 128);
  const int grid = a.ntiles;
  if (a.K % 32 != 0) return hipErrorInvalidValue;
  // Few tiles (< 2 per CU: short utterances, single clips): occupancy cannot hide the staging
  // phases, the in-wave pipelined variant wins (M = 2008: 62 -> 72 TF); many tiles: three
  // phase-separated workgroups per CU win (K = 3072: 285 vs 275 TF).  GAM_PIPE=0/1 forces one.
  static int pipe_env = -2;
  if (pipe_env == -2) { const char* e = getenv("GAM_PIPE"); pipe_env = e ? atoi(e) : -1; }
  const bool pipe = pipe_env >= 0 ? pipe_env != 0 : grid <= 512;
#define GAM_L16(ACTV)                                                                     \
  if (pipe) gam_launch_gemm16_t<ACTV, 128, true>(a, grid, stream);                        \
  else gam_launch_gemm16_t<ACTV, 128, false>(a, grid, stream);
  switch (act) {
    case GAM_ACT_SILU: GAM_L16(GAM_ACT_SILU); break;
    case GAM_ACT_RELU: GAM_L16(GAM_ACT_RELU); break;
    default: GAM_L16(GAM_ACT_NONE); break;
  }
#undef GAM_L16
  return hipGetLastError();
}
