// gam_decode_cluster.h -- RNN-T greedy decode with a CLUSTER of workgroups per utterance.
//   reference: gigaam/decoding.py:128-207 (loop), decoder.py:85-102 (predict), :41-47 (joint)
//
// The one-workgroup-per-utterance kernel (gam_decode.h) is bound by what ONE CU can pull out of L2 per
// step: W_hh (4H x H fp32 = 1.6 MB), W_pred (0.4 MB) and, for SentencePiece vocabularies, W_out
// (V x JH = 1.3 MB) -- 65 us (V = 34) to 150 us (V = 1025) per emitted symbol, on 32 of 256 CUs.
// Here C workgroups (C = 7 at 32 utterances: 224 of 256 CUs) share one utterance:
//   * member c owns H/C hidden units (all four gate rows of each, so the cell update is local), JH/C rows of
//     W_pred and V/C classes of W_out: every weight byte is read by exactly one CU per step, 16 bytes per lane
//     ([k/4][row][4] re-layout built at gam_finalize);
//   * three all-gathers per emission inside the cluster -- h' (H floats), W_pred.h' (JH floats) and the
//     per-frame (max, argmax, sum-exp) of each member's class slice -- and one per all-blank window.  The
//     control flow (frame pointer, symbol count, label) is replicated: every member takes the same decisions
//     from the same gathered values, so no control messages exist.
//   * hand-off: data-tagged 8-byte granules {f32 payload, u32 tag} written and polled with relaxed agent-scope
//     atomics (sc1, L1-bypassing; MI355X_MICROARCH.md "handoff-1to1": ~1 us).  The tag is the utterance's
//     exchange counter, buffers alternate by parity, so a granule is never rewritten before every member has
//     read it (an all-gather cannot complete before all members have written, i.e. finished the previous one).
//   * every spin is bounded (wall clock): a member that gives up sets the launch's status word, its cluster
//     unwinds and leaves counts[b] = -1 -- a workgroup that was not resident cannot hang the GPU.  The launcher sizes
//     the grid to <= one workgroup per CU, and gam_rnnt_greedy enqueues a repair pass behind every cluster launch (the
//     one-workgroup kernel with only_failed: its workgroups return at once unless their utterance was left at -1), so
//     the caller gets the reference's ids either way -- no host round trip, nothing raises.
// Tried and dropped (r02_s6): W_pred by COLUMN slices, each member publishing the partial product of its own h' slice so
// that the h' gather and the pp reduction share one hand-off -- C x JH granules per step instead of JH made it slower
// (config 3 decode 4.57 vs 4.13 ms).
// fp32 throughout; the 16-frame joint window is the same v_mfma_f32_16x16x4_f32 product as in the single-workgroup
// kernel, with reassociated sums (resident kernel: four k mod 4 partials per gate row as two v_pk_fma_f32 chains; joint
// tile: four MFMA chains), i.e. logits differ from that kernel at the 1e-6 level -- both are tested against the
// reference's log-probs (1e-3).  Inside a round only the hand-offs leave the CU: W_hh rows in registers, W_pred rows,
// the W_out slice (char vocabularies) and the next window's encoder-projection rows (global_load_lds) in LDS.
#pragma once
#include "gam_decode.h"

struct GamRnntClusterArgs {
  GamRnntArgs a;
  const float* whh_q;        // [H/4][4H][4]   W_hh: element (row r, k) at ((k/4)*4H + r)*4 + k%4
  const float* wpred_q;      // [H/4][JH][4]   joint.pred weight, same re-layout
  unsigned long long* xbuf;  // per utterance: 2 x (H + JH + C*48) granules
  int* status;               // != 0: a hand-off timed out somewhere in this launch
  int C;                     // workgroups per utterance
  int wout_slice_in_lds;     // this member's class slice of W_out is cached in LDS
  int wpred_slice_in_lds;    // this member's rows of W_pred are cached in LDS ([H/4][nP][4]): the step's pp no longer waits on L2
  int force_dead;            // test hook (GAM_RNNT_FORCE_TIMEOUT=1): odd utterances report a failed hand-off without decoding
  int* audit;                // -DGAM_RC_AUDIT=1 builds only: [0] = entries logged, entry k at [16 + 12 k] (null in production)
};

#define GAM_RC_WIN 16
// -DGAM_RC_TIMING=1: per-phase wall-clock totals (10 ns ticks) of utterance 0 / member 0 go to status[1..10] and
// gam_rnnt_greedy prints them (GAM_RNNT_TIMING=1); production builds carry none of it.
#ifndef GAM_RC_TIMING
#define GAM_RC_TIMING 0
#endif
enum { T_GATES = 1, T_XH, T_PRED, T_XP, T_Z, T_JOINT, T_XA, T_COMB, T_CTRL, T_ROUNDS };
#if GAM_RC_TIMING
#define GAM_RC_MARK(id) { const long long now_ = wall_clock64(); tacc[id] += now_ - tlast; tlast = now_; }
#else
#define GAM_RC_MARK(id)
#endif
#define GAM_RC_TIMEOUT_TICKS 100000000LL   // wall_clock64 ticks (100 MHz): 1 s
// "every load issued so far has landed", as a compiler barrier too: keeps a batch of independent loads TOGETHER in front of
// their uses (hipcc otherwise sinks each load into the conditional block that consumes it: one L2 round trip per load)
#define GAM_RC_LOADS_LANDED() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// r06 diagnosis (-DGAM_RC_AUDIT=1, never a production build): every phase of the round is computed TWICE by the same thread from freshly
// re-read inputs and every piece of LDS-held state is checked against a register copy kept by the thread that wrote it; any difference is
// logged as (site, utterance, member, exchange counter, thread, index, four words).  Sites: 1 gate row recomputed differs (words: first,
// second, xor of weight bits xor, xor of h bits xor)   2 gates[] read back != written   3 cell update recomputed differs   4 W_pred row
// recomputed differs   5 committed h / c in LDS != the writer's register copy (index 0 h, 1 c)   6 candidate h' in LDS != register copy
// 7 pp in LDS != register copy   8 window row in LDS (LDS-DMA) != the same row loaded from global   9 joint tile recomputed differs
// 10 gate_tab row re-loaded != the prefetched one
#ifndef GAM_RC_AUDIT
#define GAM_RC_AUDIT 0
#endif
#define GAM_RC_AUDIT_CAP 4096
#if GAM_RC_AUDIT
__device__ __forceinline__ void gam_rc_audit_log(int* au, int site, int b, int cm, unsigned xc, int tid, int idx, float x0, float x1, unsigned x2, unsigned x3) {
  const int k = atomicAdd(au, 1);
  if (k < GAM_RC_AUDIT_CAP) {
    int* e = au + 16 + 12 * k;
    e[0] = site; e[1] = b; e[2] = cm; e[3] = (int)xc; e[4] = tid; e[5] = idx;
    e[6] = __float_as_int(x0); e[7] = __float_as_int(x1); e[8] = (int)x2; e[9] = (int)x3;
  }
}
#define GAM_RC_LOG(site, idx, x0, x1, x2, x3) gam_rc_audit_log(g.audit, site, b, cm, xc, tid, idx, x0, x1, x2, x3)
#endif

__device__ __forceinline__ void gam_rc_put(unsigned long long* p, float v, unsigned tag) {
  const unsigned long long w = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
  __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// poll granule p0 and (when ``two``) p1 until their tags match, both loads in flight: one L2 round trip instead of two
// for the threads that own two granules of a 320-wide exchange; false on timeout / launch-wide abort
__device__ __forceinline__ bool gam_rc_get2(const unsigned long long* p0, const unsigned long long* p1, bool two, unsigned tag,
                                            float& v0, float& v1, int* status) {
  long long t_end = 0;
  bool d0 = false, d1 = !two;
  for (unsigned spin = 0;; ++spin) {
    const unsigned long long g0 = __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long g1 = __hip_atomic_load(two ? p1 : p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!d0 && (unsigned)(g0 >> 32) == tag) { v0 = __uint_as_float((unsigned)g0); d0 = true; }
    if (!d1 && (unsigned)(g1 >> 32) == tag) { v1 = __uint_as_float((unsigned)g1); d1 = true; }
    if (d0 && d1) return true;
    if ((spin & 255u) == 255u) {
      const long long now = wall_clock64();
      if (t_end == 0) t_end = now + GAM_RC_TIMEOUT_TICKS;
      if (now > t_end || __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        atomicOr(status, 1);
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

static inline size_t gam_rnnt_cluster_smem(int H, int JH, int V, int C, int nr, int wout_slice_in_lds, int wpred_slice_in_lds = 0) {
  const int nV = ((V + C - 1) / C + 15) / 16 * 16;
  size_t f = (size_t)4 * H + 256 * (size_t)nr + JH + 1024 + (size_t)GAM_RC_WIN * (JH + 4) + (size_t)GAM_RC_WIN * (nV + 1) + 64 +
             (size_t)C * 48 + 16 + (size_t)GAM_RC_WIN * JH;
  if (wout_slice_in_lds) f += (size_t)nV * (JH + 4);
  if (wpred_slice_in_lds) f += (size_t)H * ((JH + C - 1) / C);
  return sizeof(float) * f;
}
__host__ __device__ static inline size_t gam_rnnt_cluster_xgranules(int H, int JH, int C) {
  return 2 * ((size_t)H + (size_t)JH + (size_t)C * 48);
}

// NR: gate-row slots per thread, 4 * ceil(H / C) <= 256 * NR.  RESQ > 0 (only with NR == 1): H == 4 * RESQ and the
// thread keeps its gate row of W_hh (RESQ x 16 bytes) in registers for the whole decode -- at C >= 5 and H = 320 that is
// 320 VGPRs of the 512 a one-wave-per-SIMD workgroup owns, and the LSTM step stops touching L2 altogether.
// HC > 0: pred_hidden == joint_hidden == HC at compile time (every published RNN-T head: 320) -- the sizes, strides and trip counts become
// immediates instead of ~250 scalar registers' worth of spilled loop state (r06: .sgpr_spill_count 363-383 -> see profiles/r06_decode_sgpr.txt)
template <int NR, int RESQ = 0, int HC = 0>
__global__ __launch_bounds__(256) void gam_rnnt_cluster_kernel(GamRnntClusterArgs g) {
  extern __shared__ __attribute__((aligned(16))) float gam_smem_rc[];
  const GamRnntArgs& a = g.a;
  const int C = g.C;
  // block -> (utterance, member): the members of a cluster sit on one XCD (block b runs on XCD b % 8), for
  // speed only -- the hand-off is agent-scope and placement-independent
  const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
  const int b = (slot / C) * 8 + xcd, cm = slot % C;
  if (b >= a.B) return;   // (whole clusters drop out together)
  // (the resident kernel is launched only for H == JH == 4 RESQ: compile-time sizes free ~100 scalar registers)
  const int H = RESQ > 0 ? 4 * RESQ : (HC > 0 ? HC : a.H), JH = RESQ > 0 ? 4 * RESQ : (HC > 0 ? HC : a.JH), V = a.V, blank = a.V - 1;
  const int nI = (H + C - 1) / C, i0 = cm * nI, i1 = i0 + nI < H ? i0 + nI : H;        // my hidden units
  const int nP = (JH + C - 1) / C, r0 = cm * nP, r1 = r0 + nP < JH ? r0 + nP : JH;     // my rows of W_pred
  const int nV = ((V + C - 1) / C + 15) / 16 * 16, v0 = cm * nV, v1 = v0 + nV < V ? v0 + nV : V;   // my classes
  const int ZLD = JH + 4, LLD = nV + 1, WLD = JH + 4;

  float* h_s = gam_smem_rc;               // committed h (all H)
  float* hn_s = h_s + H;                  // candidate h' (all H)
  float* c_s = hn_s + H;                  // committed c of my units [nI]
  float* cn_s = c_s + H;                  // candidate c' of my units
  float* gates = cn_s + H;                // [256 * NR] my gate rows, slot = gate * nI + unit
  float* pp = gates + 256 * NR;           // W_pred.h' + b_pred (all JH)
  float* red = pp + JH;                   // [1024] partial sums of the W_pred slice
  float* zw = red + 1024;                 // [WIN][ZLD]  relu(enc + pred)   (offset is a multiple of 4 floats)
  float* lgw = zw + GAM_RC_WIN * ZLD;     // [WIN][LLD]  logits of my classes
  int* lab_s = reinterpret_cast<int*>(lgw + GAM_RC_WIN * LLD);   // [WIN]
  float* lse_s = reinterpret_cast<float*>(lab_s + GAM_RC_WIN);   // [WIN]
  int* dead_s = reinterpret_cast<int*>(lse_s + GAM_RC_WIN);      // [1] (+ padding to 64)
  float* apart = lse_s + GAM_RC_WIN + 32;                        // [C][WIN][3]  (max, argmax bits, sum-exp)
  float* zenc = apart + C * 48;                                  // [WIN][JH] encoder-projection rows of the window (LDS-DMA target)
  float* wout_l = g.wout_slice_in_lds ? zenc + GAM_RC_WIN * JH : nullptr;   // [nV][WLD] (every segment above is a multiple of 16 bytes)
  float* wpl = g.wpred_slice_in_lds ? zenc + GAM_RC_WIN * JH + (g.wout_slice_in_lds ? nV * WLD : 0) : nullptr;   // [H/4][nP][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg4 = lane >> 4;
  int len = a.enc_len[b];
  len = len < 0 ? 0 : (len > a.Tp ? a.Tp : len);

  unsigned long long* xh = g.xbuf + (size_t)b * gam_rnnt_cluster_xgranules(H, JH, C);   // [2][H]
  unsigned long long* xp = xh + 2 * H;                                                  // [2][JH]
  unsigned long long* xa = xp + 2 * JH;                                                 // [2][C][WIN][3]

  for (int i = tid; i < H; i += 256) { h_s[i] = 0.f; c_s[i] = 0.f; }
  if (tid == 0) dead_s[0] = 0;
  if (wout_l != nullptr)
    for (int i = tid; i < (v1 > v0 ? v1 - v0 : 0) * JH; i += 256) wout_l[(i / JH) * WLD + (i % JH)] = a.wout[(size_t)v0 * JH + i];
  if (wpl != nullptr)
    for (int i = tid; i < (H / 4) * nP; i += 256) {
      const int q = i / nP, rr = i - q * nP;
      const int r = r0 + rr < JH ? r0 + rr : JH - 1;
      *reinterpret_cast<f32x4*>(wpl + (size_t)i * 4) = *reinterpret_cast<const f32x4*>(g.wpred_q + ((size_t)q * JH + r) * 4);
    }
  __syncthreads();

  // my gate-row slots: slot s = tid + 256 j -> gate s / nI, unit i0 + s % nI
  int grow[NR];
  bool gok[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int s = tid + 256 * j, gi = s / nI, ii = s - gi * nI;
    gok[j] = gi < 4 && i0 + ii < i1;
    grow[j] = gok[j] ? gi * H + i0 + ii : 0;
  }
  // W_pred slice: P k-parts per row when the slice is small
  const int P = nP >= 128 ? 1 : (256 / nP < 8 ? 256 / nP : 8);
  const int HQ = H / 4, JQ = JH / 4;
  f32x4 wres[RESQ > 0 ? RESQ : 1];
  if constexpr (RESQ > 0) {
#pragma unroll
    for (int q = 0; q < RESQ; ++q) wres[q] = gam_rc_glb4(g.whh_q + ((size_t)q * 4 * H + grow[0]) * 4);
  }

  // encoder-projection rows of the 16-frame window starting at frame tw, straight into LDS (global_load_lds_dwordx4:
  // a wave moves 64 x 16 bytes to 1 KiB of zenc, no registers held): in flight while the round that decided tw
  // finishes and the next predictor step runs; JH % 16 == 0 (launcher) makes the window a whole number of wave loads
  auto fetch_window = [&](int tw) {
    const int Wn = len - tw < GAM_RC_WIN ? len - tw : GAM_RC_WIN;
    for (int c = wave; c < GAM_RC_WIN * JQ / 64; c += 4) {
      const int idx = c * 64 + lane;
      const int f = idx / JQ, q = idx - f * JQ;
      int tt = tw + (f < Wn ? f : Wn - 1);
      tt = tt < 0 ? 0 : (tt < a.Tp ? tt : a.Tp - 1);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.encp + ((size_t)b * a.Tp + tt) * JH + 4 * q),
                                       (__attribute__((address_space(3))) void*)(zenc + c * 256), 16, 0, 0);
    }
  };
  fetch_window(0);
  int label = V;       // gate_tab row V: zero embedding (predict(None, None), decoder.py:97-100)
  float tabv[NR];      // gate_tab[label] of my rows: fetched as soon as the label is known (end of the round that emitted it)
#pragma unroll
  for (int j = 0; j < NR; ++j) tabv[j] = gam_rc_glb1(a.gate_tab + (size_t)label * 4 * H + grow[j]);
  int n_out = 0, n_dump = 0;
  int t = 0, sym = 0;
  bool need_pred = true;
  unsigned xc = 0;     // exchange counter = tag
#if GAM_RC_AUDIT
  float au_h[2] = {0.f, 0.f}, au_c[2] = {0.f, 0.f}, au_hn[2] = {0.f, 0.f}, au_cn[2] = {0.f, 0.f}, au_pp[2] = {0.f, 0.f};   // (nI, nP <= 512)
#endif
  int par_h = 0, par_p = 0, par_a = 0;
  bool dead = g.force_dead && (b & 1);

#if GAM_RC_TIMING
  long long tacc[12] = {0}, tlast = wall_clock64();
#endif
  while (t < len && !dead) {
#if GAM_RC_TIMING
    tacc[T_ROUNDS] += 1;
#endif
    if (need_pred) {
#if GAM_RC_AUDIT
      {
        int k = 0;
        for (int ii = tid; i0 + ii < i1; ii += 256, ++k) {
          if (__float_as_uint(h_s[i0 + ii]) != __float_as_uint(au_h[k])) GAM_RC_LOG(5, 2 * ii, h_s[i0 + ii], au_h[k], 0u, 0u);
          if (__float_as_uint(c_s[ii]) != __float_as_uint(au_c[k])) GAM_RC_LOG(5, 2 * ii + 1, c_s[ii], au_c[k], 0u, 0u);
        }
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          const float tv = a.gate_tab[(size_t)label * 4 * H + grow[j]];
          if (__float_as_uint(tv) != __float_as_uint(tabv[j])) GAM_RC_LOG(10, j, tv, tabv[j], (unsigned)label, 0u);
        }
      }
#endif
      // ---- LSTM gates of my units: tab[label] + W_hh.h, k ascending (same fmaf chain as the 1-workgroup kernel)
      {
        float acc[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[j] = tabv[j];
        if constexpr (RESQ > 0) {
          // four partial sums (k mod 4) as two packed-fp32 chains: 2 x RESQ v_pk_fma_f32 instead of 4 x RESQ dependent fmas
          f32x2 pa = (f32x2){acc[0], 0.f}, pb = (f32x2){0.f, 0.f};
#pragma unroll
          for (int q = 0; q < RESQ; ++q) {
            const f32x4 hv = *reinterpret_cast<const f32x4*>(h_s + 4 * q);
            pa = __builtin_elementwise_fma((f32x2){wres[q].x, wres[q].y}, (f32x2){hv.x, hv.y}, pa);
            pb = __builtin_elementwise_fma((f32x2){wres[q].z, wres[q].w}, (f32x2){hv.z, hv.w}, pb);
          }
          acc[0] = (pa.x + pa.y) + (pb.x + pb.y);
        } else {
        // float4 loads in flight per row slot: the step is L2-LATENCY bound (each batch of loads is one round trip),
        // so as many as the registers hold -- ~160 VGPRs of weights per thread
        constexpr int KU = NR == 1 ? 40 : (NR == 2 ? 20 : (NR == 3 ? 12 : (NR == 5 ? 8 : 4)));
#pragma unroll 1
        for (int q0 = 0; q0 < HQ; q0 += KU) {   // (one batch of loads per trip also when H is a compile-time constant)
          f32x4 w[KU][NR];
#pragma unroll
          for (int u = 0; u < KU; ++u) {
            const int q = q0 + u < HQ ? q0 + u : HQ - 1;
#pragma unroll
            for (int j = 0; j < NR; ++j) w[u][j] = gam_rc_glb4(g.whh_q + ((size_t)q * 4 * H + grow[j]) * 4);
          }
#pragma unroll
          for (int u = 0; u < KU; ++u) {
            if (q0 + u < HQ) {
              const f32x4 hv = *reinterpret_cast<const f32x4*>(h_s + 4 * (q0 + u));
#pragma unroll
              for (int j = 0; j < NR; ++j) {
                acc[j] = fmaf(w[u][j].x, hv.x, acc[j]);
                acc[j] = fmaf(w[u][j].y, hv.y, acc[j]);
                acc[j] = fmaf(w[u][j].z, hv.z, acc[j]);
                acc[j] = fmaf(w[u][j].w, hv.w, acc[j]);
              }
            }
          }
        }
        }
#if GAM_RC_AUDIT
        if constexpr (RESQ == 0) {   // second pass: the same sums from re-loaded weights and re-read h
          float acc2[NR];
          unsigned wx[2] = {0u, 0u}, hx[2] = {0u, 0u};
#pragma unroll
          for (int j = 0; j < NR; ++j) acc2[j] = tabv[j];
          constexpr int KU2 = NR == 1 ? 40 : (NR == 2 ? 20 : (NR == 3 ? 12 : (NR == 5 ? 8 : 4)));
          for (int pass = 0; pass < 2; ++pass) {   // pass 0 recomputes; pass 1 only re-reads (a third look at the same words)
            for (int q0 = 0; q0 < HQ; q0 += KU2) {
              f32x4 w[KU2][NR];
#pragma unroll
              for (int u = 0; u < KU2; ++u) {
                const int q = q0 + u < HQ ? q0 + u : HQ - 1;
#pragma unroll
                for (int j = 0; j < NR; ++j) w[u][j] = gam_rc_glb4(g.whh_q + ((size_t)q * 4 * H + grow[j]) * 4);
              }
#pragma unroll
              for (int u = 0; u < KU2; ++u) {
                if (q0 + u < HQ) {
                  const f32x4 hv = *reinterpret_cast<const f32x4*>(h_s + 4 * (q0 + u));
                  hx[pass] ^= __float_as_uint(hv.x) ^ (__float_as_uint(hv.y) * 3u) ^ (__float_as_uint(hv.z) * 5u) ^ (__float_as_uint(hv.w) * 7u);
#pragma unroll
                  for (int j = 0; j < NR; ++j) {
                    wx[pass] ^= (__float_as_uint(w[u][j].x) ^ (__float_as_uint(w[u][j].y) * 3u) ^ (__float_as_uint(w[u][j].z) * 5u) ^ (__float_as_uint(w[u][j].w) * 7u)) * (unsigned)(2 * j + 1);
                    if (pass == 0) {
                      acc2[j] = fmaf(w[u][j].x, hv.x, acc2[j]);
                      acc2[j] = fmaf(w[u][j].y, hv.y, acc2[j]);
                      acc2[j] = fmaf(w[u][j].z, hv.z, acc2[j]);
                      acc2[j] = fmaf(w[u][j].w, hv.w, acc2[j]);
                    }
                  }
                }
              }
            }
          }
#pragma unroll
          for (int j = 0; j < NR; ++j)
            if (gok[j] && (__float_as_uint(acc2[j]) != __float_as_uint(acc[j]) || wx[0] != wx[1] || hx[0] != hx[1]))
              GAM_RC_LOG(1, j, acc[j], acc2[j], wx[0] ^ wx[1], hx[0] ^ hx[1]);
        }
#endif
#pragma unroll
        for (int j = 0; j < NR; ++j)
          if (gok[j]) gates[tid + 256 * j] = acc[j];
#if GAM_RC_AUDIT
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NR; ++j)
          if (gok[j] && __float_as_uint(gates[tid + 256 * j]) != __float_as_uint(acc[j])) GAM_RC_LOG(2, j, gates[tid + 256 * j], acc[j], 0u, 0u);
#endif
      }
      __syncthreads();
    GAM_RC_MARK(T_GATES);
      ++xc;
      for (int ii = tid; i0 + ii < i1; ii += 256) {   // cell update of my units (gate order i, f, g, o)
        const float ig = gam_sigmoid_exact(gates[ii]), fg = gam_sigmoid_exact(gates[nI + ii]);
        const float gg = tanhf(gates[2 * nI + ii]), og = gam_sigmoid_exact(gates[3 * nI + ii]);
        const float cn = fg * c_s[ii] + ig * gg;
        const float hn = og * tanhf(cn);
        cn_s[ii] = cn;
#if GAM_RC_AUDIT
        {
          const volatile float* gv = gates;
          const volatile float* cv = c_s;
          const float ig2 = gam_sigmoid_exact(gv[ii]), fg2 = gam_sigmoid_exact(gv[nI + ii]);
          const float gg2 = tanhf(gv[2 * nI + ii]), og2 = gam_sigmoid_exact(gv[3 * nI + ii]);
          const float cn2 = fg2 * cv[ii] + ig2 * gg2;
          const float hn2 = og2 * tanhf(cn2);
          if (__float_as_uint(cn2) != __float_as_uint(cn) || __float_as_uint(hn2) != __float_as_uint(hn)) GAM_RC_LOG(3, ii, hn, hn2, __float_as_uint(cn), __float_as_uint(cn2));
          au_hn[ii >> 8] = hn; au_cn[ii >> 8] = cn;
        }
#endif
        if (C > 1) gam_rc_put(xh + par_h * H + i0 + ii, hn, xc);
        else hn_s[i0 + ii] = hn;
      }
      if (C > 1) {
        for (int i = tid; i < H; i += 512) {
          float va = 0.f, vb = 0.f;
          const bool two = i + 256 < H;
          if (!gam_rc_get2(xh + par_h * H + i, xh + par_h * H + i + 256, two, xc, va, vb, g.status)) { dead_s[0] = 1; va = vb = 0.f; }
          hn_s[i] = va;
          if (two) hn_s[i + 256] = vb;
        }
        par_h ^= 1;
      }
      __syncthreads();
    GAM_RC_MARK(T_XH);
#if GAM_RC_AUDIT
      {
        int k = 0;
        for (int ii = tid; i0 + ii < i1; ii += 256, ++k)
          if (__float_as_uint(hn_s[i0 + ii]) != __float_as_uint(au_hn[k])) GAM_RC_LOG(6, ii, hn_s[i0 + ii], au_hn[k], 0u, 0u);
      }
#endif
      if (dead_s[0]) { dead = true; break; }
      // ---- my rows of W_pred.h' + b_pred
      {
        if (P == 1) {
          for (int rr = tid; r0 + rr < r1; rr += 256) {
            const int r = r0 + rr;
            float acc = gam_rc_glb1(a.bpred + r);
            for (int q0 = 0; q0 < HQ; q0 += 16) {
              f32x4 w[16];
#pragma unroll
              for (int u = 0; u < 16; ++u) {
                const int q = q0 + u < HQ ? q0 + u : HQ - 1;
                if (wpl != nullptr) w[u] = gam_rc_lds4(wpl + ((size_t)q * nP + rr) * 4);
                else w[u] = gam_rc_glb4(g.wpred_q + ((size_t)q * JH + r) * 4);
              }
#pragma unroll
              for (int u = 0; u < 16; ++u)
                if (q0 + u < HQ) {
                  const f32x4 hv = *reinterpret_cast<const f32x4*>(hn_s + 4 * (q0 + u));
                  acc = fmaf(w[u].x, hv.x, acc); acc = fmaf(w[u].y, hv.y, acc); acc = fmaf(w[u].z, hv.z, acc); acc = fmaf(w[u].w, hv.w, acc);
                }
            }
            red[rr] = acc;
#if GAM_RC_AUDIT
            {
              float acc2 = a.bpred[r];
              for (int q0 = 0; q0 < HQ; q0 += 16) {
                f32x4 w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                  const int q = q0 + u < HQ ? q0 + u : HQ - 1;
                  if (wpl != nullptr) w[u] = gam_rc_lds4(wpl + ((size_t)q * nP + rr) * 4);
                  else w[u] = gam_rc_glb4(g.wpred_q + ((size_t)q * JH + r) * 4);
                }
#pragma unroll
                for (int u = 0; u < 16; ++u)
                  if (q0 + u < HQ) {
                    const f32x4 hv = *reinterpret_cast<const f32x4*>(hn_s + 4 * (q0 + u));
                    acc2 = fmaf(w[u].x, hv.x, acc2); acc2 = fmaf(w[u].y, hv.y, acc2); acc2 = fmaf(w[u].z, hv.z, acc2); acc2 = fmaf(w[u].w, hv.w, acc2);
                  }
              }
              if (__float_as_uint(acc2) != __float_as_uint(acc)) GAM_RC_LOG(4, rr, acc, acc2, 0u, 0u);
            }
#endif
          }
        } else {
          const int rr = tid % nP, part = tid / nP;
          if (part < P && r0 + rr < r1) {
            const int r = r0 + rr;
            const int qa = part * HQ / P, qb = (part + 1) * HQ / P;
            float acc = part == 0 ? gam_rc_glb1(a.bpred + r) : 0.f;
            for (int q0 = qa; q0 < qb; q0 += 16) {
              f32x4 w[16];
#pragma unroll
              for (int u = 0; u < 16; ++u) {
                const int q = q0 + u < qb ? q0 + u : qb - 1;
                if (wpl != nullptr) w[u] = gam_rc_lds4(wpl + ((size_t)q * nP + rr) * 4);
                else w[u] = gam_rc_glb4(g.wpred_q + ((size_t)q * JH + r) * 4);
              }
#pragma unroll
              for (int u = 0; u < 16; ++u)
                if (q0 + u < qb) {
                  const f32x4 hv = *reinterpret_cast<const f32x4*>(hn_s + 4 * (q0 + u));
                  acc = fmaf(w[u].x, hv.x, acc); acc = fmaf(w[u].y, hv.y, acc); acc = fmaf(w[u].z, hv.z, acc); acc = fmaf(w[u].w, hv.w, acc);
                }
            }
            red[part * nP + rr] = acc;
          }
        }
      }
      __syncthreads();
    GAM_RC_MARK(T_PRED);
      ++xc;
      for (int rr = tid; r0 + rr < r1; rr += 256) {
        float v = red[rr];
        for (int p = 1; p < P; ++p) v += red[p * nP + rr];
        if (C > 1) gam_rc_put(xp + par_p * JH + r0 + rr, v, xc);
        else pp[r0 + rr] = v;
#if GAM_RC_AUDIT
        au_pp[rr >> 8] = v;
#endif
      }
      if (C > 1) {
        for (int i = tid; i < JH; i += 512) {
          float va = 0.f, vb = 0.f;
          const bool two = i + 256 < JH;
          if (!gam_rc_get2(xp + par_p * JH + i, xp + par_p * JH + i + 256, two, xc, va, vb, g.status)) { dead_s[0] = 1; va = vb = 0.f; }
          pp[i] = va;
          if (two) pp[i + 256] = vb;
        }
        par_p ^= 1;
      }
      need_pred = false;
      __syncthreads();
    GAM_RC_MARK(T_XP);
      if (dead_s[0]) { dead = true; break; }
    }

    // ---- joint of frames t .. t+W-1 with the current predictor state: z = relu(enc + pred) (all JH, every member)
    const int W = len - t < GAM_RC_WIN ? len - t : GAM_RC_WIN;
    // (the window's encoder-projection rows were fetched when t was decided, at the end of the previous round)
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0): my wave's LDS-DMA loads have landed ...
    __syncthreads();                      // ... and so have everyone else's
#if GAM_RC_AUDIT
    {
      int k = 0;
      for (int rr = tid; r0 + rr < r1; rr += 256, ++k)
        if (__float_as_uint(pp[r0 + rr]) != __float_as_uint(au_pp[k])) GAM_RC_LOG(7, rr, pp[r0 + rr], au_pp[k], 0u, 0u);
      const int Wn = len - t < GAM_RC_WIN ? len - t : GAM_RC_WIN;
      for (int idx = tid; idx < GAM_RC_WIN * JQ; idx += 256) {
        const int f = idx / JQ, q = idx - f * JQ;
        int tt = t + (f < Wn ? f : Wn - 1);
        tt = tt < 0 ? 0 : (tt < a.Tp ? tt : a.Tp - 1);
        const f32x4 ev = *reinterpret_cast<const f32x4*>(zenc + 4 * idx);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(a.encp + ((size_t)b * a.Tp + tt) * JH + 4 * q);
        if (__float_as_uint(ev.x) != __float_as_uint(gv.x) || __float_as_uint(ev.y) != __float_as_uint(gv.y) || __float_as_uint(ev.z) != __float_as_uint(gv.z) ||
            __float_as_uint(ev.w) != __float_as_uint(gv.w))
          GAM_RC_LOG(8, idx, ev.x, gv.x, (unsigned)t, (unsigned)tt);
      }
    }
#endif
    for (int idx = tid; idx < GAM_RC_WIN * JQ; idx += 256) {
      const int f = idx / JQ, q = idx - f * JQ;
      const f32x4 ev = *reinterpret_cast<const f32x4*>(zenc + 4 * idx);
      const f32x4 pv = *reinterpret_cast<const f32x4*>(pp + 4 * q);
      *reinterpret_cast<f32x4*>(zw + f * ZLD + 4 * q) =
          (f32x4){fmaxf(ev.x + pv.x, 0.f), fmaxf(ev.y + pv.y, 0.f), fmaxf(ev.z + pv.z, 0.f), fmaxf(ev.w + pv.w, 0.f)};
    }
    __syncthreads();
    GAM_RC_MARK(T_Z);
    // logits[f][v] = bout[v] + sum_k z[f][k] wout[v][k] for my classes: one 16x16 MFMA tile per 16 classes
    for (int nt = wave; v0 + nt * 16 < v1; nt += 4) {
      const int v = v0 + nt * 16 + li;
      const int vc = v < V ? v : V - 1;
      // The W_out slice lives either in LDS (char vocabularies at C >= 3) or in global memory: two instantiations of the loop
      // with address-space-qualified loads (gam_rc_lds4 / gam_rc_glb4) -- one generic pointer selected at run time made every
      // load a FLAT instruction (r05: the library now holds none).
      const float* zr = zw + li * ZLD + 4 * lg4;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc, acc2 = acc, acc3 = acc;   // (k mod 4 chains: MFMA latency, not rate, bounds one tile)
      // all of the tile's W_out reads in flight at once (JH / 16 x 16 bytes per lane): with the slice streamed from L2
      // (SentencePiece vocabularies) every batch of loads is one L2 round trip, and a wave runs several tiles
      constexpr int MAXU = GAM_RNNT_MAXH / 16, UB = RESQ > 0 ? 10 : 32;   // (register-resident W_hh leaves room for 10 at a time)
      auto tile = [&](auto load4, const float* wr, const bool from_global) {
#pragma unroll
        for (int u0 = 0; u0 < MAXU; u0 += UB) {
          if (16 * u0 + 16 > JH) break;
          f32x4 wf[UB];
#pragma unroll
          for (int u = 0; u < UB; ++u) wf[u] = load4(wr + (16 * (u0 + u) + 16 <= JH ? 16 * (u0 + u) : 0));
          if (from_global) GAM_RC_LOADS_LANDED();      // the whole batch in flight, ONE round trip
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            if (16 * (u0 + u) + 16 <= JH) {
              const f32x4 zf = gam_rc_lds4(zr + 16 * (u0 + u));
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zf.x, wf[u].x, acc, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(zf.y, wf[u].y, acc1, 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(zf.z, wf[u].z, acc2, 0, 0, 0);
              acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(zf.w, wf[u].w, acc3, 0, 0, 0);
            }
          }
        }
      };
      if (wout_l != nullptr) tile([](const float* p) { return gam_rc_lds4(p); }, wout_l + (size_t)(vc - v0) * WLD + 4 * lg4, false);
      else tile([](const float* p) { return gam_rc_glb4(p); }, a.wout + (size_t)vc * JH + 4 * lg4, true);
      acc = (acc + acc1) + (acc2 + acc3);
#if GAM_RC_AUDIT
      {
        const f32x4 first = acc;
        acc = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1 = acc; acc2 = acc; acc3 = acc;
        if (wout_l != nullptr) tile([](const float* p) { return gam_rc_lds4(p); }, wout_l + (size_t)(vc - v0) * WLD + 4 * lg4, false);
        else tile([](const float* p) { return gam_rc_glb4(p); }, a.wout + (size_t)vc * JH + 4 * lg4, true);
        acc = (acc + acc1) + (acc2 + acc3);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (__float_as_uint(first[r]) != __float_as_uint(acc[r])) GAM_RC_LOG(9, nt * 64 + lane, first[r], acc[r], (unsigned)r, 0u);
      }
#endif
      if (v < v1) {   // C/D: col = lane&15 = class, row = 4*(lane>>4) + r = frame
        const float bo = gam_rc_glb1(a.bout + v);
#pragma unroll
        for (int r = 0; r < 4; ++r) lgw[(4 * lg4 + r) * LLD + (v - v0)] = acc[r] + bo;
      }
    }
    __syncthreads();
    GAM_RC_MARK(T_JOINT);
    // per-frame (max, first argmax, sum-exp) over my classes: 16 lanes per frame, wave w takes frames 4w .. 4w+3
    ++xc;
    const int NG = a.dump != nullptr ? 3 : 2;   // (the sum-exp is only exchanged when log-probs are dumped)
    {
      const int f = 4 * wave + lg4;
      const float* lr = lgw + f * LLD;
      float best = -INFINITY;
      int bi = 0x7fffffff;
      if (f < W)
        for (int vl = li; v0 + vl < v1; vl += 16) {
          const float x = lr[vl];
          if (x > best) { best = x; bi = v0 + vl; }   // (ascending vl per lane: the first maximum stays)
        }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      float se = 0.f;
      if (a.dump != nullptr) {
        if (f < W)
          for (int vl = li; v0 + vl < v1; vl += 16) se += expf(lr[vl] - best);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
      }
      if (li == 0) {
        if (C > 1) {
          unsigned long long* q = xa + (((size_t)par_a * C + cm) * GAM_RC_WIN + f) * 3;
          gam_rc_put(q + 0, best, xc);
          gam_rc_put(q + 1, __int_as_float(bi), xc);
          if (NG == 3) gam_rc_put(q + 2, se, xc);
        } else {
          apart[f * 3 + 0] = best; apart[f * 3 + 1] = __int_as_float(bi); apart[f * 3 + 2] = se;
        }
      }
    }
    if (C > 1) {
      const int n = C * GAM_RC_WIN * NG;
      for (int i = tid; i < n; i += 512) {
        float va = 0.f, vb = 0.f;
        const bool two = i + 256 < n;
        const int ia = i / NG * 3 + i % NG, ib = (i + 256) / NG * 3 + (i + 256) % NG;
        const unsigned long long* base = xa + (size_t)par_a * C * GAM_RC_WIN * 3;
        if (!gam_rc_get2(base + ia, base + ib, two, xc, va, vb, g.status)) { dead_s[0] = 1; va = vb = 0.f; }
        apart[ia] = va;
        if (two) apart[ib] = vb;
      }
      par_a ^= 1;
    }
    __syncthreads();
    GAM_RC_MARK(T_XA);
    if (dead_s[0]) { dead = true; break; }
    // combine the members' slices (ascending class order: the first maximum wins).  Every wave does it for all 16
    // frames (lane & 15 = frame), so the window's verdict needs no further barrier: a ballot gives the first
    // non-blank frame, a lane read its label
    int fstar, kstar;
    {
      const int f = li;
      float M = -INFINITY;
      int bi = 0x7fffffff;
      for (int c = 0; c < C; ++c) {
        const float m = apart[(c * GAM_RC_WIN + f) * 3 + 0];
        const int ix = __float_as_int(apart[(c * GAM_RC_WIN + f) * 3 + 1]);
        if (m > M || (m == M && ix < bi)) { M = m; bi = ix; }
      }
      const unsigned nb = (unsigned)(__ballot(f < W && bi != blank) & 0xffffull);
      fstar = nb != 0u ? __builtin_ctz(nb) : W;
      kstar = __shfl(bi, fstar & 15, 64);
      if (a.dump != nullptr) {
        float S = 0.f;
        for (int c = 0; c < C; ++c) {
          const float m = apart[(c * GAM_RC_WIN + f) * 3 + 0];
          if (m > -INFINITY) S += apart[(c * GAM_RC_WIN + f) * 3 + 2] * expf(m - M);
        }
        if (tid < GAM_RC_WIN) lse_s[f] = M + logf(S);
        __syncthreads();
      }
    }
    GAM_RC_MARK(T_COMB);
    const int n_eval = fstar < W ? fstar + 1 : W;   // joint evaluations the sequential loop performs
    if (a.dump != nullptr) {
      for (int f = 0; f < n_eval; ++f) {
        if (n_dump + f < a.dump_cap) {
          float* dp = a.dump + ((size_t)b * a.dump_cap + n_dump + f) * V;
          const float lse = lse_s[f];
          for (int vl = tid; v0 + vl < v1; vl += 256) dp[v0 + vl] = lgw[f * LLD + vl] - lse;
        }
      }
    }
    n_dump += n_eval;
    if (fstar == W) {            // W blank frames
      t += W;
      sym = 0;
    } else {                     // emission at frame t + fstar (decoding.py:175-178)
      const int k = kstar;
      const int te = t + fstar;
      if (fstar > 0) sym = 0;
      if (cm == 0 && tid == 0 && n_out < a.cap) {
        a.ids[(size_t)b * a.cap + n_out] = k;
        a.frames[(size_t)b * a.cap + n_out] = te;
      }
      ++n_out;
      ++sym;
      label = k;
#pragma unroll
      for (int j = 0; j < NR; ++j) tabv[j] = gam_rc_glb1(a.gate_tab + (size_t)label * 4 * H + grow[j]);   // lands while the round finishes
      { float* x = h_s; h_s = hn_s; hn_s = x; x = c_s; c_s = cn_s; cn_s = x; }   // commit (h', c'): swap the buffers
#if GAM_RC_AUDIT
      au_h[0] = au_hn[0]; au_h[1] = au_hn[1]; au_c[0] = au_cn[0]; au_c[1] = au_cn[1];
#endif
      need_pred = true;
      if (sym >= a.max_symbols) { t = te + 1; sym = 0; }   // frame advances regardless (decoding.py:189-205)
      else t = te;
    }
    if (t < len) fetch_window(t);
    __syncthreads();
    GAM_RC_MARK(T_CTRL);
  }
#if GAM_RC_TIMING
  if (b == 0 && cm == 0 && tid == 0)
    for (int i = 1; i <= T_ROUNDS; ++i) g.status[i] = (int)tacc[i];
#endif
  if (cm == 0 && tid == 0) {
    a.counts[b] = dead ? -1 : (n_out < a.cap ? n_out : a.cap);
    if (a.dump_count != nullptr) a.dump_count[b] = n_dump;
  }
}
