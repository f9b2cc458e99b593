// gam_attn16.h -- the fused attention of gam_attn.h with both small GEMMs evaluated as
// three-term fp16 splits on the fp16 matrix cores (see gam_gemm16.h for the numerics):
//   S^T = K.Q^T        ~  K_hi.Q_hi + K_hi.Q_lo + K_lo.Q_hi
//   O^T = V^T.P^T      ~  V_hi.P_hi + V_hi.P_lo + V_lo.P_hi        (fp32 accumulate)
// d_k = 48 is contracted as one v_mfma_f32_16x16x32_f16 (d 0..31) + one
// v_mfma_f32_16x16x16_f16 (d 32..47); keys are contracted 32 at a time, the k-slot order
// (e < 4: key block 2c, e >= 4: key block 2c+1) chosen so that a lane's eight S^T
// accumulator registers of two neighbouring key blocks ARE its B operand -- P still never
// leaves the registers.  Softmax statistics, masking and the rescale stay fp32.
// K and V^T tiles are split once while they are staged into LDS (fp16 planes, rows padded to
// an odd number of 16-byte slots).  Same work decomposition as gam_attn_f32_kernel.
#pragma once
#include "gam_attn.h"
#include "gam_gemm16.h"

#define GAM_A16_KLD 56   // halfs per K-plane row  (112 B = 7 x 16 B)
#define GAM_A16_VLD 72   // halfs per V^T-plane row (144 B = 9 x 16 B)

__device__ __forceinline__ void gam_split8(const float (&v)[8], gam_half8& hi, gam_half8& lo) {
  gam_u32x4 h, l;   // (gam_common.h gam_split2: 4 VALU instructions per pair)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned hh, ll;
    gam_split2(v[2 * i], v[2 * i + 1], hh, ll);
    h[i] = hh;
    l[i] = ll;
  }
  hi = __builtin_bit_cast(gam_half8, h);
  lo = __builtin_bit_cast(gam_half8, l);
}

// REL: the relative-position scores of v1 models (gam_attn.h, reference encoder.py:191-228):
//   S[i,j] = ((q_i + u).k_j + (q_i + v).P(i - j)) / sqrt(d_k); the second term is evaluated per key
// tile as G[m][query] = P(rlo + m).(q + v) for the 80 relative positions the (16 queries x 64 keys)
// block touches -- same three-term split, P rows split on the fly -- and skewed into S^T through LDS.
#ifndef GAM_ATT_NJ
#define GAM_ATT_NJ 2   // 16-query sub-blocks per wave at large grids (r02 A/B: 1 = 64 queries per workgroup was 192 vs 148 us at 32 x 20 s)
#endif
// NJ is a template parameter since r04: small grids (a few utterances per GPU) take NJ = 1 -- 64 queries per workgroup, twice the
// workgroups -- when 128-query workgroups would leave the chip under one workgroup per CU (gam_launch_attn_mode).
// TERMS = 3: the three-term split above (fp32-equivalent).  TERMS = 1 (GAM_GEMM_F16, the opt-in speed mode): hi planes only --
// K, V, Q and P are rounded to fp16 once, one MFMA per product, fp32 accumulation and fp32 softmax statistics: the arithmetic of
// an fp16 flash attention (the reference's GPU default runs SDPA under fp16 autocast, /root/reference/gigaam/model.py:34-37).
// The lo planes are neither computed nor stored (the compiler drops the dead halves of the splits).
// -DGAM_ATT_PV_TERMS=2 (r06 EXPERIMENT build; never the product): the probabilities enter P.V as fp16 only (the V_hi.P_lo
// product and the lo half of the P split are dropped: 1/3 of the P.V MFMAs, 2 of the ~10 VALU instructions per score).  Results in
// profiles/r06_attn_pv2.txt.
#ifndef GAM_ATT_PV_TERMS
#define GAM_ATT_PV_TERMS 3
#endif
template <bool REL, int TERMS = 3, int NJ = GAM_ATT_NJ>
__global__ __launch_bounds__(256, NJ == 1 ? (REL ? 3 : 4) : 2) void gam_attn_f16x3_kernel(GamAttnArgs a) {
  constexpr bool LO = TERMS == 3;
  a.scale *= 1.44269504088896341f;   // softmax via 2^x: p = 2^(s*log2e - m)
  __shared__ float Gs[REL ? 4 * 80 * 17 : 1];   // per wave: (q+v).P for 80 relative positions x 16 queries
  __shared__ __attribute__((aligned(16))) _Float16 Kh[GAM_ATT_KT * GAM_A16_KLD];
  __shared__ __attribute__((aligned(16))) _Float16 Kl[GAM_ATT_KT * GAM_A16_KLD];
  __shared__ __attribute__((aligned(16))) _Float16 Vh[GAM_ATT_DK * GAM_A16_VLD];
  __shared__ __attribute__((aligned(16))) _Float16 Vl[GAM_ATT_DK * GAM_A16_VLD];
  constexpr int DK = GAM_ATT_DK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  int klen = a.Tv;
  if (a.lens != nullptr) { const int l = a.lens[b]; klen = l < a.Tv ? l : a.Tv; }
  const GamRows ur = gam_rows(a.cu, b, a.Ta, klen);   // (packed rows: only the utterance's own frames exist)
  if ((int)blockIdx.x * (64 * NJ) >= ur.lim) return;  // a query block behind the last frame (whole workgroup: no barrier is left behind)
  const size_t rowbase = ur.base;
  const int rlim = ur.lim;
  const int qw0 = blockIdx.x * (64 * NJ) + wave * (16 * NJ);

  // Q fragments (B operand of S^T), pre-scaled: x32 part d = 8*lg .. +7, x16 part d = 32 + 4*lg .. +3
  gam_half8 qh32[NJ], ql32[NJ];
  gam_half4 qh16[NJ], ql16[NJ];
  gam_half8 gh32[NJ], gl32[NJ];   // REL: (q + pos_bias_v) * scale
  gam_half4 gh16[NJ], gl16[NJ];
  int qrow[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int qi = qw0 + j * 16 + li;
    qrow[j] = qi;
    const int qc = qi < rlim ? qi : rlim - 1;
    const float* qp = a.q + (rowbase + qc) * a.ldq + h * DK;
    const float4 q0 = *reinterpret_cast<const float4*>(qp + 8 * lg);
    const float4 q1 = *reinterpret_cast<const float4*>(qp + 8 * lg + 4);
    const float4 q2 = *reinterpret_cast<const float4*>(qp + 32 + 4 * lg);
    float v8[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    float v4[4] = {q2.x, q2.y, q2.z, q2.w};
    if (REL) {
      const float* pu = a.pos_u + h * DK;
      const float* pv = a.pos_v + h * DK;
      float g8[8], g4[4];
#pragma unroll
      for (int e = 0; e < 8; ++e) { g8[e] = (v8[e] + pv[8 * lg + e]) * a.scale; v8[e] += pu[8 * lg + e]; }
#pragma unroll
      for (int e = 0; e < 4; ++e) { g4[e] = (v4[e] + pv[32 + 4 * lg + e]) * a.scale; v4[e] += pu[32 + 4 * lg + e]; }
      gam_split8(g8, gh32[j], gl32[j]);
      gam_split4((f32x4){g4[0], g4[1], g4[2], g4[3]}, gh16[j], gl16[j]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v8[e] *= a.scale;
    gam_split8(v8, qh32[j], ql32[j]);
    gam_split4((f32x4){v4[0] * a.scale, v4[1] * a.scale, v4[2] * a.scale, v4[3] * a.scale}, qh16[j], ql16[j]);
  }

  f32x4 o[3][NJ];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int j = 0; j < NJ; ++j) o[d][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float mrun[NJ], lsum[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) { mrun[j] = -INFINITY; lsum[j] = 0.f; }

  // K / V rows of a key tile: fetched into registers one tile AHEAD (in flight under the previous tile's MFMAs),
  // split and written to LDS at the top of their own iteration
  float4 kreg[3], vreg[2][2];
  auto fetch_tile = [&](int kt0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int idx = tid + i * 256;       // 0..767
      const int kr = idx / 12, c4 = (idx - kr * 12) * 4;
      int key = kt0 + kr;
      key = key < rlim ? key : rlim - 1;
      kreg[i] = *reinterpret_cast<const float4*>(a.k + (rowbase + key) * a.ldq + h * DK + c4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int it = tid + i * 256;
      it = it < 32 * 12 ? it : 32 * 12 - 1;
      const int kp = it / 12, c4 = (it - kp * 12) * 4;
      int k0 = kt0 + 2 * kp, k1 = k0 + 1;
      k0 = k0 < rlim ? k0 : rlim - 1;
      k1 = k1 < rlim ? k1 : rlim - 1;
      vreg[i][0] = *reinterpret_cast<const float4*>(a.v + (rowbase + k0) * a.ldv + h * DK + c4);
      vreg[i][1] = *reinterpret_cast<const float4*>(a.v + (rowbase + k1) * a.ldv + h * DK + c4);
    }
  };
  if (!REL && klen > 0) fetch_tile(0);

  for (int kt0 = 0; kt0 < klen; kt0 += GAM_ATT_KT) {
    if constexpr (REL) {   // (no look-ahead: the REL fragments leave no registers for it)
      // ---- stage K [64][48] and V^T [48][64] as fp16 hi/lo planes ----
  #pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int idx = tid + i * 256;       // 0..767
        const int kr = idx / 12, c4 = (idx - kr * 12) * 4;
        int key = kt0 + kr;
        key = key < rlim ? key : rlim - 1;
        const float4 kv = *reinterpret_cast<const float4*>(a.k + (rowbase + key) * a.ldq + h * DK + c4);
        gam_half4 hi, lo;
        gam_split4((f32x4){kv.x, kv.y, kv.z, kv.w}, hi, lo);
        *reinterpret_cast<gam_half4*>(&Kh[kr * GAM_A16_KLD + c4]) = hi;
        if (LO) *reinterpret_cast<gam_half4*>(&Kl[kr * GAM_A16_KLD + c4]) = lo;
      }
      // V^T: one item = 2 neighbouring keys x 4 channels -> per channel one 4-byte write per plane
      for (int it = tid; it < 32 * 12; it += 256) {
        const int kp = it / 12, c4 = (it - kp * 12) * 4;
        int k0 = kt0 + 2 * kp, k1 = k0 + 1;
        k0 = k0 < rlim ? k0 : rlim - 1;
        k1 = k1 < rlim ? k1 : rlim - 1;
        const float4 v0 = *reinterpret_cast<const float4*>(a.v + (rowbase + k0) * a.ldv + h * DK + c4);
        const float4 v1 = *reinterpret_cast<const float4*>(a.v + (rowbase + k1) * a.ldv + h * DK + c4);
        gam_half4 h0, l0, h1, l1;
        gam_split4((f32x4){v0.x, v0.y, v0.z, v0.w}, h0, l0);
        gam_split4((f32x4){v1.x, v1.y, v1.z, v1.w}, h1, l1);
        typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
  #pragma unroll
        for (int e = 0; e < 4; ++e) {
          *reinterpret_cast<half2_t*>(&Vh[(c4 + e) * GAM_A16_VLD + 2 * kp]) = (half2_t){h0[e], h1[e]};
          if (LO) *reinterpret_cast<half2_t*>(&Vl[(c4 + e) * GAM_A16_VLD + 2 * kp]) = (half2_t){l0[e], l1[e]};
        }
      }
    } else {
      // ---- stage K [64][48] and V^T [48][64] as fp16 hi/lo planes ----
  #pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int idx = tid + i * 256;
        const int kr = idx / 12, c4 = (idx - kr * 12) * 4;
        gam_half4 hi, lo;
        gam_split4((f32x4){kreg[i].x, kreg[i].y, kreg[i].z, kreg[i].w}, hi, lo);
        *reinterpret_cast<gam_half4*>(&Kh[kr * GAM_A16_KLD + c4]) = hi;
        if (LO) *reinterpret_cast<gam_half4*>(&Kl[kr * GAM_A16_KLD + c4]) = lo;
      }
      // V^T: one item = 2 neighbouring keys x 4 channels -> per channel one 4-byte write per plane
  #pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int it = tid + i * 256;
        if (it < 32 * 12) {
          const int kp = it / 12, c4 = (it - kp * 12) * 4;
          gam_half4 h0, l0, h1, l1;
          gam_split4((f32x4){vreg[i][0].x, vreg[i][0].y, vreg[i][0].z, vreg[i][0].w}, h0, l0);
          gam_split4((f32x4){vreg[i][1].x, vreg[i][1].y, vreg[i][1].z, vreg[i][1].w}, h1, l1);
          typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
  #pragma unroll
          for (int e = 0; e < 4; ++e) {
            *reinterpret_cast<half2_t*>(&Vh[(c4 + e) * GAM_A16_VLD + 2 * kp]) = (half2_t){h0[e], h1[e]};
            if (LO) *reinterpret_cast<half2_t*>(&Vl[(c4 + e) * GAM_A16_VLD + 2 * kp]) = (half2_t){l0[e], l1[e]};
          }
        }
      }
    }
    __syncthreads();
    if (!REL && kt0 + GAM_ATT_KT < klen) fetch_tile(kt0 + GAM_ATT_KT);

    // ---- S^T[kb][j] = K_kb . Q_j^T  (6 MFMAs per 16x16 tile) ----
    f32x4 st[4][NJ];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int ko = (kb * 16 + li) * GAM_A16_KLD;
      const gam_half8 kh32 = *reinterpret_cast<const gam_half8*>(&Kh[ko + 8 * lg]);
      const gam_half4 kh16 = *reinterpret_cast<const gam_half4*>(&Kh[ko + 32 + 4 * lg]);
      gam_half8 kl32 = kh32;
      gam_half4 kl16 = kh16;
      if (LO) {
        kl32 = *reinterpret_cast<const gam_half8*>(&Kl[ko + 8 * lg]);
        kl16 = *reinterpret_cast<const gam_half4*>(&Kl[ko + 32 + 4 * lg]);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        // Two accumulator chains, one per instruction shape: a 4-pass 16x16x16 MFMA issued
        // right behind an 8-pass 16x16x32 one ON THE SAME accumulator lost the 8-pass result
        // (hipcc 7.2 / gfx950, observed: S came out without its K_hi.Q_hi term).  Same-shape
        // dependent chains are safe (the GEMMs rely on them); the two sums are added on the VALU.
        f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 s16 = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (LO) {
          s = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl32, qh32[j], s, 0, 0, 0);
          s16 = __builtin_amdgcn_mfma_f32_16x16x16f16(kl16, qh16[j], s16, 0, 0, 0);
          s = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh32, ql32[j], s, 0, 0, 0);
          s16 = __builtin_amdgcn_mfma_f32_16x16x16f16(kh16, ql16[j], s16, 0, 0, 0);
        }
        s = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh32, qh32[j], s, 0, 0, 0);
        s16 = __builtin_amdgcn_mfma_f32_16x16x16f16(kh16, qh16[j], s16, 0, 0, 0);
        s += s16;
        st[kb][j] = s;
      }
    }

    if (REL) {
      // ---- S^T[key a][query b] += G[b - a + 63][b],  G[m][b] = P(rlo + m) . (q_b + v) ----
      float* gw = Gs + wave * (80 * 17);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int rlo = (qw0 + j * 16) - kt0 - 63 + (a.Tv - 1);   // pbuf row of m = 0
#pragma unroll
        for (int mt = 0; mt < 5; ++mt) {
          int prow = rlo + mt * 16 + li;
          prow = prow < 0 ? 0 : (prow > 2 * a.Tv - 2 ? 2 * a.Tv - 2 : prow);
          const float* pp = a.pbuf + (size_t)prow * a.ldp + h * DK;
          const float4 p0 = *reinterpret_cast<const float4*>(pp + 8 * lg);
          const float4 p1 = *reinterpret_cast<const float4*>(pp + 8 * lg + 4);
          const float4 p2 = *reinterpret_cast<const float4*>(pp + 32 + 4 * lg);
          const float p8[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
          gam_half8 ph32, pl32;
          gam_half4 ph16, pl16;
          gam_split8(p8, ph32, pl32);
          gam_split4((f32x4){p2.x, p2.y, p2.z, p2.w}, ph16, pl16);
          f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f};
          f32x4 g16 = (f32x4){0.f, 0.f, 0.f, 0.f};   // one chain per MFMA shape (see below)
          if (LO) {
            g = __builtin_amdgcn_mfma_f32_16x16x32_f16(pl32, gh32[j], g, 0, 0, 0);
            g16 = __builtin_amdgcn_mfma_f32_16x16x16f16(pl16, gh16[j], g16, 0, 0, 0);
            g = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph32, gl32[j], g, 0, 0, 0);
            g16 = __builtin_amdgcn_mfma_f32_16x16x16f16(ph16, gl16[j], g16, 0, 0, 0);
          }
          g = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph32, gh32[j], g, 0, 0, 0);
          g16 = __builtin_amdgcn_mfma_f32_16x16x16f16(ph16, gh16[j], g16, 0, 0, 0);
          g += g16;
#pragma unroll
          for (int r = 0; r < 4; ++r) gw[(mt * 16 + lg * 4 + r) * 17 + li] = g[r];
        }
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) st[kb][j][r] += gw[(li - (kb * 16 + lg * 4 + r) + 63) * 17 + li];
        __syncthreads();
      }
    }

    // ---- key mask + online softmax (fp32, per query = per lane column).  Only the LAST key tile of an utterance can hold
    //      keys >= klen: every other tile skips the compare + select per score (2 of the ~10 VALU instructions a score
    //      costs in this VALU-bound kernel); the branch is workgroup-uniform.
    const bool full_tile = kt0 + GAM_ATT_KT <= klen;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      float mx = -INFINITY;
      if (full_tile) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kb][j][r]);
      } else {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kt0 + kb * 16 + lg * 4 + r;
            float s = st[kb][j][r];
            s = key < klen ? s : -INFINITY;
            st[kb][j][r] = s;
            mx = fmaxf(mx, s);
          }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mnew = fmaxf(mrun[j], mx);   // finite: key kt0 < klen is in this tile
      const float alpha = __builtin_amdgcn_exp2f(mrun[j] - mnew);
      mrun[j] = mnew;
      float ps = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(st[kb][j][r] - mnew);
          st[kb][j][r] = p;
          ps += p;
        }
      lsum[j] = lsum[j] * alpha + ps;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        o[d][j][0] *= alpha; o[d][j][1] *= alpha;
        o[d][j][2] *= alpha; o[d][j][3] *= alpha;
      }
    }

    // ---- O^T[d][j] += V^T_d . P_j^T over 32-key chunks: slot (lg, e) = key (2c + (e>>2))*16 + 4*lg + (e&3) ----
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      gam_half8 ph[NJ], pl[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const float p8[8] = {st[2 * c][j][0], st[2 * c][j][1], st[2 * c][j][2], st[2 * c][j][3],
                             st[2 * c + 1][j][0], st[2 * c + 1][j][1], st[2 * c + 1][j][2], st[2 * c + 1][j][3]};
        gam_split8(p8, ph[j], pl[j]);
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const int vo = (d * 16 + li) * GAM_A16_VLD + 4 * lg;
        const gam_half4 vh0 = *reinterpret_cast<const gam_half4*>(&Vh[vo + (2 * c) * 16]);
        const gam_half4 vh1 = *reinterpret_cast<const gam_half4*>(&Vh[vo + (2 * c + 1) * 16]);
        const gam_half8 vh = {vh0[0], vh0[1], vh0[2], vh0[3], vh1[0], vh1[1], vh1[2], vh1[3]};
        gam_half8 vl = vh;
        if (LO) {
          const gam_half4 vl0 = *reinterpret_cast<const gam_half4*>(&Vl[vo + (2 * c) * 16]);
          const gam_half4 vl1 = *reinterpret_cast<const gam_half4*>(&Vl[vo + (2 * c + 1) * 16]);
          vl = (gam_half8){vl0[0], vl0[1], vl0[2], vl0[3], vl1[0], vl1[1], vl1[2], vl1[3]};
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (LO) {
            o[d][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph[j], o[d][j], 0, 0, 0);
            if (GAM_ATT_PV_TERMS == 3) o[d][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl[j], o[d][j], 0, 0, 0);
          }
          o[d][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph[j], o[d][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue: lane (query li, group lg) holds O^T rows d = dt*16 + 4*lg + r ----
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    float l = lsum[j];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = l > 0.f ? 1.0f / l : 0.f;   // klen == 0 -> zeros
    if (qrow[j] < rlim) {
#pragma unroll
      for (int d = 0; d < 3; ++d)
        gam_store4(a.ctx, (size_t)(rowbase + qrow[j]) * a.ldo, h * DK + 4 * lg + 16 * d, o[d][j][0] * inv, o[d][j][1] * inv,
                   o[d][j][2] * inv, o[d][j][3] * inv, a.ctx_split);
    }
  }
}

// split = true: fp16-split MFMA path; false: exact-fp32 MFMA path (gam_attn.h)
// terms = 3: the three-term split (GAM_GEMM_F16X3); terms = 1: hi planes only (GAM_GEMM_F16, opt-in)
static inline hipError_t gam_launch_attn_mode(const GamAttnArgs& a, int dk, bool split, hipStream_t s, int terms = 3, int ncu = 256) {
  if (!split) return gam_launch_attn(a, dk, s);
  if (dk != GAM_ATT_DK) return hipErrorInvalidValue;
  static const int nj_env = []() { const char* e = getenv("GAM_ATT_NJ_SMALL"); return e ? atoi(e) : -1; }();   // A/B switch (0: never NJ = 1)
  const long wgs128 = (long)gam_cdiv(a.Ta, 64 * GAM_ATT_NJ) * a.H * a.B;
  // (measured, same box: one clip 3.10 -> 3.05 ms, 2 x 20 s 4.69 -> 4.64; at 4 x 20 s -- 256 workgroups of 128 queries -- nothing: threshold = one per CU)
  const bool small = GAM_ATT_NJ == 2 && (nj_env < 0 ? wgs128 < (long)ncu : (nj_env > 0 && wgs128 < (long)nj_env * ncu));
  const bool rel = a.pbuf != nullptr;
#define GAM_ATT_GO(REL_, T_, NJ_)                                                                      \
  hipLaunchKernelGGL((gam_attn_f16x3_kernel<REL_, T_, NJ_>), dim3(gam_cdiv(a.Ta, 64 * NJ_), a.H, a.B), dim3(256), 0, s, a)
  if (small) {
    if (terms == 1) { if (rel) GAM_ATT_GO(true, 1, 1); else GAM_ATT_GO(false, 1, 1); }
    else { if (rel) GAM_ATT_GO(true, 3, 1); else GAM_ATT_GO(false, 3, 1); }
  } else {
    if (terms == 1) { if (rel) GAM_ATT_GO(true, 1, GAM_ATT_NJ); else GAM_ATT_GO(false, 1, GAM_ATT_NJ); }
    else { if (rel) GAM_ATT_GO(true, 3, GAM_ATT_NJ); else GAM_ATT_GO(false, 3, GAM_ATT_NJ); }
  }
#undef GAM_ATT_GO
  return hipGetLastError();
}
