// gam_attn.h -- fused multi-head self-attention, fp32, LDS-staged K / V^T tiles,
// online softmax, v_mfma_f32_16x16x4_f32 for both small GEMMs.
//
// Reference: gigaam/encoder.py:265-274 (rotary models: F.scaled_dot_product_attention
// with key-padding mask, scale 1/sqrt(d_k)) and :159-188.  Keys t >= len_b are excluded
// (att_mask, encoder.py:616-624); at batch 1 the reference passes no mask at all.
//
// Work decomposition: grid (q-tiles of 128, heads, batch); 4 waves per workgroup, each
// wave owns 32 query rows (2 blocks of 16).  Scores are computed TRANSPOSED,
// S^T = K . Q^T, so that in the MFMA C/D layout (col = lane&15 = query, row = key) a
// lane's 4 accumulator registers are 4 consecutive keys of ONE query: (a) the row
// softmax statistics reduce in-lane + 2 xor-shuffles, (b) the registers feed the second
// MFMA (O^T = V^T . P^T) directly as its B operand -- P never moves through LDS -- and
// (c) O^T has the same query-per-lane layout, so the online-softmax rescale is a plain
// per-lane multiply and the epilogue stores 16 B per lane.
#pragma once
#include "gam_common.h"
#include "gam_pack.h"

#define GAM_ATT_DK 48
#define GAM_ATT_KT 64          // keys per LDS tile
#define GAM_ATT_KLD 52         // K tile row stride (floats): 13 x 16 B
#define GAM_ATT_VLD 68         // V^T tile row stride (floats)

struct GamAttnArgs {
  const float* q;   // [B*Ta, ldq] rows, head h at column h*dk
  const float* k;   // same row stride
  const float* v;   // [B*Ta, ldv]
  float* ctx;       // [B*Ta, ldo]
  int ctx_split;    // ctx in the sp32 GEMM-operand layout (ldo % 32 == 0)
  const int* lens;  // valid frames per utterance (keys), or null = no mask
  const int* cu;    // packed rows (gam_pack.h): first row of every utterance; null = padded layout (b * Ta)
  int B, Ta, Tv, H;
  long ldq, ldv, ldo;
  float scale;      // 1/sqrt(d_k); the kernels work in the log2 domain (scale * log2 e, v_exp_f32)
  // relative-position variant (v1 models, reference encoder.py:191-228):
  //   scores[i,j] = ((q_i + u).k_j + (q_i + v).P(i - j)) / sqrt(d_k),  P(r) = W_pos.pe(r)
  const float* pbuf;   // [2*Tv-1, ldp]: row n <-> relative position n - (Tv-1); head h at column h*dk
  const float* pos_u;  // [H*dk]
  const float* pos_v;  // [H*dk]
  long ldp;
};

template <bool REL>
__global__ __launch_bounds__(256) void gam_attn_f32_kernel(GamAttnArgs a) {
  a.scale *= 1.44269504088896341f;   // softmax via 2^x: p = 2^(s*log2e - m)
  __shared__ __attribute__((aligned(16))) float Ks[GAM_ATT_KT * GAM_ATT_KLD];
  __shared__ __attribute__((aligned(16))) float Vt[GAM_ATT_DK * GAM_ATT_VLD];
  __shared__ float Gs[REL ? 4 * 80 * 17 : 1];   // per wave: (q+v).P for 80 relative positions x 16 queries
  constexpr int DK = GAM_ATT_DK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  int klen = a.Tv;
  if (a.lens != nullptr) { const int l = a.lens[b]; klen = l < a.Tv ? l : a.Tv; }
  const GamRows ur = gam_rows(a.cu, b, a.Ta, klen);
  if ((int)blockIdx.x * 128 >= ur.lim) return;
  const size_t rowbase = ur.base;
  const int rlim = ur.lim;
  const int qw0 = blockIdx.x * 128 + wave * 32;

  // Q fragments (B operand of S^T): lane (query li, kk lg) holds d = 16*s + 4*lg + e
  float4 qf[2][3];
  float4 qvf[2][3];   // REL only: (q + pos_bias_v) * scale
  int qrow[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int qi = qw0 + j * 16 + li;
    qrow[j] = qi;
    const int qc = qi < rlim ? qi : rlim - 1;
    const float* qp = a.q + (rowbase + qc) * a.ldq + h * DK + 4 * lg;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      float4 t = *reinterpret_cast<const float4*>(qp + 16 * s);
      if (REL) {
        const float4 u = *reinterpret_cast<const float4*>(a.pos_u + h * DK + 4 * lg + 16 * s);
        const float4 v = *reinterpret_cast<const float4*>(a.pos_v + h * DK + 4 * lg + 16 * s);
        qvf[j][s] = make_float4((t.x + v.x) * a.scale, (t.y + v.y) * a.scale, (t.z + v.z) * a.scale, (t.w + v.w) * a.scale);
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      t.x *= a.scale; t.y *= a.scale; t.z *= a.scale; t.w *= a.scale;
      qf[j][s] = t;
    }
  }

  f32x4 o[3][2];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int j = 0; j < 2; ++j) o[d][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float mrun[2] = {-INFINITY, -INFINITY};
  float lsum[2] = {0.f, 0.f};

  for (int kt0 = 0; kt0 < klen; kt0 += GAM_ATT_KT) {
    // ---- stage K [64][48] and V^T [48][64] ----
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int idx = tid + i * 256;       // 0..767
      const int kr = idx / 12, c4 = (idx - kr * 12) * 4;
      int key = kt0 + kr;
      key = key < rlim ? key : rlim - 1;
      const float4 kv = *reinterpret_cast<const float4*>(a.k + (rowbase + key) * a.ldq + h * DK + c4);
      const float4 vv = *reinterpret_cast<const float4*>(a.v + (rowbase + key) * a.ldv + h * DK + c4);
      *reinterpret_cast<float4*>(&Ks[kr * GAM_ATT_KLD + c4]) = kv;
      Vt[(c4 + 0) * GAM_ATT_VLD + kr] = vv.x;
      Vt[(c4 + 1) * GAM_ATT_VLD + kr] = vv.y;
      Vt[(c4 + 2) * GAM_ATT_VLD + kr] = vv.z;
      Vt[(c4 + 3) * GAM_ATT_VLD + kr] = vv.w;
    }
    __syncthreads();

    // ---- S^T[kb][j] = K_kb . Q_j^T ----
    f32x4 st[4][2];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      st[kb][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      st[kb][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* kp = &Ks[(kb * 16 + li) * GAM_ATT_KLD + 4 * lg];
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const float4 kf = *reinterpret_cast<const float4*>(kp + 16 * s);
        st[kb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[0][s].x, st[kb][0], 0, 0, 0);
        st[kb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[1][s].x, st[kb][1], 0, 0, 0);
        st[kb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[0][s].y, st[kb][0], 0, 0, 0);
        st[kb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[1][s].y, st[kb][1], 0, 0, 0);
        st[kb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[0][s].z, st[kb][0], 0, 0, 0);
        st[kb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[1][s].z, st[kb][1], 0, 0, 0);
        st[kb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[0][s].w, st[kb][0], 0, 0, 0);
        st[kb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[1][s].w, st[kb][1], 0, 0, 0);
      }
    }

    if (REL) {
      // ---- S^T[key a][query b] += G[b - a + 63][b],  G[m][b] = P(rlo + m) . (q_b + v) ----
      float* gw = Gs + wave * (80 * 17);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int rlo = (qw0 + j * 16) - kt0 - 63 + (a.Tv - 1);   // pbuf row of m = 0
#pragma unroll
        for (int mt = 0; mt < 5; ++mt) {
          int prow = rlo + mt * 16 + li;
          prow = prow < 0 ? 0 : (prow > 2 * a.Tv - 2 ? 2 * a.Tv - 2 : prow);
          const float* pp = a.pbuf + (size_t)prow * a.ldp + h * DK + 4 * lg;
          f32x4 gacc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            const float4 pf = *reinterpret_cast<const float4*>(pp + 16 * s);
            gacc = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.x, qvf[j][s].x, gacc, 0, 0, 0);
            gacc = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.y, qvf[j][s].y, gacc, 0, 0, 0);
            gacc = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.z, qvf[j][s].z, gacc, 0, 0, 0);
            gacc = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.w, qvf[j][s].w, gacc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) gw[(mt * 16 + lg * 4 + r) * 17 + li] = gacc[r];
        }
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) st[kb][j][r] += gw[(li - (kb * 16 + lg * 4 + r) + 63) * 17 + li];
        __syncthreads();
      }
    }

    // ---- key mask + online softmax (per query = per lane column) ----
    float alpha[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float mx = -INFINITY;
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
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mnew = fmaxf(mrun[j], mx);   // finite: key kt0 < klen is in this tile
      alpha[j] = __builtin_amdgcn_exp2f(mrun[j] - mnew);
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
      lsum[j] = lsum[j] * alpha[j] + ps;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        o[d][j][0] *= alpha[j]; o[d][j][1] *= alpha[j];
        o[d][j][2] *= alpha[j]; o[d][j][3] *= alpha[j];
      }
    }

    // ---- O^T[d][j] += V^T_d . P_j^T : step (kb, e) contracts keys kb*16 + 4*lg + e ----
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const float4 vf = *reinterpret_cast<const float4*>(&Vt[(d * 16 + li) * GAM_ATT_VLD + kb * 16 + 4 * lg]);
        o[d][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.x, st[kb][0][0], o[d][0], 0, 0, 0);
        o[d][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.x, st[kb][1][0], o[d][1], 0, 0, 0);
        o[d][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.y, st[kb][0][1], o[d][0], 0, 0, 0);
        o[d][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.y, st[kb][1][1], o[d][1], 0, 0, 0);
        o[d][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.z, st[kb][0][2], o[d][0], 0, 0, 0);
        o[d][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.z, st[kb][1][2], o[d][1], 0, 0, 0);
        o[d][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.w, st[kb][0][3], o[d][0], 0, 0, 0);
        o[d][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.w, st[kb][1][3], o[d][1], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: lane (query li, group lg) holds O^T rows d = dt*16 + 4*lg + r ----
#pragma unroll
  for (int j = 0; j < 2; ++j) {
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

static inline hipError_t gam_launch_attn(const GamAttnArgs& a, int dk, hipStream_t s) {
  if (dk != GAM_ATT_DK) return hipErrorInvalidValue;
  dim3 grid(gam_cdiv(a.Ta, 128), a.H, a.B);
  if (a.pbuf != nullptr) hipLaunchKernelGGL(gam_attn_f32_kernel<true>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(gam_attn_f32_kernel<false>, grid, dim3(256), 0, s, a);
  return hipGetLastError();
}
