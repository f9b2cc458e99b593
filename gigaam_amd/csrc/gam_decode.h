// gam_decode.h -- greedy decoders as wavefront-level primitives, zero host syncs.
//   CTC   : reference gigaam/decoding.py:56-96  (argmax, drop blank / repeats / t >= len, compact)
//   RNN-T : reference gigaam/decoding.py:128-207 + decoder.py:41-47,85-102
#pragma once
#include "gam_common.h"

// ------------------------------------------------------------------ log_softmax rows
// one wave per row of [rows, V] (decoder.py:18-21 log_softmax over classes)
__global__ __launch_bounds__(256) void gam_log_softmax_kernel(const float* x, float* y, int rows, int V) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * V;
  float mx = -INFINITY;
  for (int v = lane; v < V; v += 64) mx = fmaxf(mx, xr[v]);
  mx = gam_wave_max(mx);
  float s = 0.f;
  for (int v = lane; v < V; v += 64) s += expf(xr[v] - mx);
  s = gam_wave_sum(s);
  const float lse = mx + logf(s);
  float* yr = y + (size_t)row * V;
  for (int v = lane; v < V; v += 64) yr[v] = xr[v] - lse;
}

// ------------------------------------------------------------------ CTC greedy
// one workgroup (GAM_CTC_NT threads) per utterance.  Phase 1: argmax with torch's first-max tie rule -> labels in LDS.
// Phase 2: keep = label != blank && (t == 0 || label != label[t-1]) && t < len; ballot/popcount prefix compaction.
// Phase 1 has two forms.  Character vocabularies (V <= 64, the published CTC models: V = 34): ONE THREAD PER FRAME -- a
// frame's V logits are V/2 eight-byte loads of one lane, all of them in flight before the first compare, the scan is a
// register loop with no cross-lane traffic, and 501 frames are one pass of the 512 threads: a single L2 round trip instead
// of the 16 dependent ones of the wave-per-frame form (r03: 71 us at every batch size -- 2.3 % of a single clip's step).
// Larger vocabularies (V = 257 / 1025): one wave per frame, lanes stride the classes.
#define GAM_CTC_NT 512
__global__ __launch_bounds__(GAM_CTC_NT) void gam_ctc_greedy_kernel(const float* logits, const int* enc_len, int Tp, int V,
                                                                    int* ids, int* frames, int* counts) {
  extern __shared__ int gam_smem_ctc[];   // labels[Tp] + wave totals[NT / 64]
  constexpr int NWV = GAM_CTC_NT / 64;
  int* lab = gam_smem_ctc;
  int* wtot = gam_smem_ctc + Tp;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int blank = V - 1;
  int len = enc_len[b];
  len = len < 0 ? 0 : (len > Tp ? Tp : len);   // decoding.py:76 clamp
  if (V <= 64 && (V & 1) == 0) {
    const int hv = V >> 1;
    for (int t = tid; t < Tp; t += GAM_CTC_NT) {
      const float2* xr = reinterpret_cast<const float2*>(logits + ((size_t)b * Tp + t) * V);   // (V even: 8-byte aligned rows)
      float2 x[32];
#pragma unroll
      for (int j = 0; j < 32; ++j) x[j] = xr[j < hv ? j : 0];
      float best = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        if (j < hv) {    // ascending class order + strict '>' = first maximum
          if (x[j].x > best || bi == 0x7fffffff) { best = x[j].x; bi = 2 * j; }
          if (x[j].y > best) { best = x[j].y; bi = 2 * j + 1; }
        }
      }
      lab[t] = bi;
    }
  } else {
    for (int t = wave; t < Tp; t += NWV) {
      const float* xr = logits + ((size_t)b * Tp + t) * V;
      float best = -INFINITY;
      int bi = 0x7fffffff;
      for (int v = lane; v < V; v += 64) {
        const float x = xr[v];
        if (x > best || (x == best && v < bi)) { best = x; bi = v; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      if (lane == 0) lab[t] = bi;
    }
  }
  __syncthreads();
  int base = 0;
  for (int t0 = 0; t0 < Tp; t0 += GAM_CTC_NT) {
    const int t = t0 + tid;
    bool keep = false;
    int l = 0;
    if (t < len) {
      l = lab[t];
      keep = (l != blank) && (t == 0 || l != lab[t - 1]);
    }
    const unsigned long long m = __ballot(keep);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wtot[wave] = __popcll(m);
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NWV; ++w) {
      const int c = wtot[w];
      woff += w < wave ? c : 0;
      tot += c;
    }
    if (keep) {
      const int pos = base + woff + before;
      ids[(size_t)b * Tp + pos] = l;
      frames[(size_t)b * Tp + pos] = t;
    }
    base += tot;
    __syncthreads();
  }
  if (tid == 0) counts[b] = base;
}

// ------------------------------------------------------------------ RNN-T greedy
// One persistent workgroup per utterance runs the whole frame/symbol loop on device, no
// host syncs (reference: >= 3 per step, decoding.py:165,171,173).
//
//  * The reference re-runs predict(last_label, state) on every step although its inputs
//    only change after a non-blank emission (decoding.py:156-160,175-178): here the
//    candidate (g, h', c') and W_pred.g are computed once per commit and reused.
//  * The encoder half of the joint (W_enc.f_t + b) is hoisted to one GEMM over all frames;
//    W_ih.embed[v] + b_ih + b_hh is a [V+1, 4H] table built at finalize.
//  * Between two emissions the predictor term is constant, so the joint of the next
//    WIN = 16 frames is evaluated at once as a [16 x JH] x [JH x V] product on the matrix
//    cores (v_mfma_f32_16x16x4_f32) and scanned for the first non-blank frame: every frame
//    before it is a confirmed blank step, everything after it is discarded and re-evaluated
//    with the new predictor state.  Same arithmetic per (frame, state) as the sequential
//    loop, ~T/16 + #tokens steps instead of T + #tokens.
struct GamRnntArgs {
  const float* encp;     // [B*Tp, JH]  W_enc.f + b_enc
  const int* enc_len;    // [B]
  const float* gate_tab; // [V+1, 4H]; row V = no-label (zero embedding)
  const float* whh_t;    // [H, 4H]   W_hh transposed
  const float* wpred_t;  // [H, JH]   joint.pred weight transposed
  const float* bpred;    // [JH]
  const float* wout;     // [V, JH]   joint_net.1 weight
  const float* bout;     // [V]
  int* ids; int* frames; int* counts;   // [B, cap], [B, cap], [B]
  float* dump; int* dump_count;         // optional [B, dump_cap, V] log-probs of every joint call
  int B, Tp, V, H, JH, max_symbols, cap, dump_cap;
  int wout_in_lds;       // set by the launcher: W_out (V x JH fp32) is cached in LDS
  int only_failed;       // repair pass behind the cluster kernel: decode only the utterances it left at counts[b] < 0
  // Predictor layers above the first (nn.LSTM(pred_hidden, pred_hidden, pred_rnn_layers), reference decoder.py:78-83): layer l
  // takes the layer below's NEW hidden state as input.  L = 1 for every published checkpoint; L > 1 runs on this kernel only.
  int L;
  const float* wih_x;    // [L-1][H][4H]  weight_ih_l{1..} transposed
  const float* whh_x;    // [L-1][H][4H]  weight_hh_l{1..} transposed
  const float* bias_x;   // [L-1][4H]     bias_ih + bias_hh
};

#define GAM_RNNT_MAXH 512
#define GAM_RNNT_RPT 8   // gate rows per thread: 4*H <= 2048
#define GAM_RNNT_MAXV 2048
#define GAM_RNNT_WIN 16

// 16-byte loads with the address space stated (the decode kernels hold no FLAT instruction: every pointer that may be LDS or global gets two instantiations of its loop)
__device__ __forceinline__ f32x4 gam_rc_lds4(const float* p) {
  return *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((__attribute__((address_space(3))) const void*)(p));
}
__device__ __forceinline__ f32x4 gam_rc_glb4(const float* p) {
  return *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>((__attribute__((address_space(1))) const void*)(p));
}
__device__ __forceinline__ float gam_rc_glb1(const float* p) { return *p; }

static inline size_t gam_rnnt_smem(int H, int JH, int V, int wout_in_lds, int L = 1) {
  const int vp = (V + 15) / 16 * 16;
  size_t f = (size_t)8 * H + 2 * JH + (size_t)GAM_RNNT_WIN * (JH + 4) + (size_t)GAM_RNNT_WIN * (vp + 1) + 64 + 16;
  if (wout_in_lds) f += (size_t)V * (JH + 4);
  f += (size_t)(L > 1 ? L - 1 : 0) * 4 * H;   // (h, c, h', c') of the predictor layers above the first
  return sizeof(float) * f;
}

template <int NR>   // gate rows per thread: 4*H <= 256*NR
__global__ __launch_bounds__(256) void gam_rnnt_greedy_kernel(GamRnntArgs a) {
  extern __shared__ __attribute__((aligned(16))) float gam_smem_rnnt[];
  const int H = a.H, JH = a.JH, V = a.V, blank = a.V - 1;
  const int VP = (V + 15) / 16 * 16, ZLD = JH + 4, LLD = VP + 1;
  float* h_s = gam_smem_rnnt;            // committed state
  float* c_s = h_s + H;
  float* hn_s = c_s + H;                 // candidate state (= g)
  float* cn_s = hn_s + H;
  float* gates = cn_s + H;               // [4H]
  float* pp = gates + 4 * H;             // W_pred.g + b_pred  [JH]
  float* zw = pp + 2 * JH;               // [WIN][ZLD]  relu(enc + pred)   (16-byte aligned: 8H + 2JH floats)
  float* lgw = zw + GAM_RNNT_WIN * ZLD;  // [WIN][LLD]  logits
  int* lab_s = reinterpret_cast<int*>(lgw + GAM_RNNT_WIN * LLD);   // [WIN]
  float* mx_s = reinterpret_cast<float*>(lab_s + GAM_RNNT_WIN);    // [WIN]
  float* lse_s = mx_s + GAM_RNNT_WIN;                              // [WIN]
  const int WLD = JH + 4;
  float* wout_l = a.wout_in_lds ? lse_s + GAM_RNNT_WIN + 12 : nullptr;   // [V][WLD], 16-byte aligned
  // state of layers 1 .. L-1, behind everything else: layer l at xs + (l - 1) * 4H as [h | c | h' | c']
  float* xs = lse_s + GAM_RNNT_WIN + 12 + (a.wout_in_lds ? (size_t)V * WLD : 0);
  const int L = a.L > 1 ? a.L : 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg4 = lane >> 4;
  const int b = blockIdx.x;
  if (a.only_failed && a.counts[b] >= 0) return;   // (block-uniform) the cluster kernel finished this utterance
  int len = a.enc_len[b];
  len = len < 0 ? 0 : (len > a.Tp ? a.Tp : len);

  for (int i = tid; i < H; i += 256) { h_s[i] = 0.f; c_s[i] = 0.f; }
  for (int i = tid; i < (L - 1) * 4 * H; i += 256) xs[i] = 0.f;
  if (wout_l != nullptr)
    for (int i = tid; i < V * JH; i += 256) wout_l[(i / JH) * WLD + (i % JH)] = a.wout[i];
  __syncthreads();

  // (frame, k) of this thread's window elements, fixed for the whole decode (index clamped: every load
  // is unconditional)
  int zfk[GAM_RNNT_WIN * GAM_RNNT_MAXH / 256];
#pragma unroll
  for (int u = 0; u < GAM_RNNT_WIN * GAM_RNNT_MAXH / 256; ++u) {
    int idx = tid + 256 * u;
    idx = idx < GAM_RNNT_WIN * JH ? idx : GAM_RNNT_WIN * JH - 1;
    const int f = idx / JH;
    zfk[u] = (f << 16) | (idx - f * JH);
  }
  int label = V;       // gate_tab row V: zero embedding (predict(None, None), decoder.py:97-100)
  int n_out = 0, n_dump = 0;
  int t = 0, sym = 0;  // current frame, symbols already emitted on it
  bool need_pred = true;

  while (t < len) {
    if (need_pred) {
      // ---- LSTM cell: gates = tab[label] + W_hh.h  (gate order i,f,g,o); thread = gate row(s),
      //      W_hh^T rows are contiguous over the gate index (coalesced), 8 k in flight ----
      {
        // 16 k x NR rows of W_hh^T in flight per thread: the step is L2-latency-bound, so the
        // number of independent loads per wait is what sets its duration.  Every load is
        // unconditional (row index clamped): a per-element "load or 0" select makes hipcc
        // branch around each load and wait for it (cdna_hip_programming.md §5 trap c).
        float acc[NR];
        int roff[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          roff[j] = tid + 256 * j < 4 * H ? tid + 256 * j : 4 * H - 1;
          acc[j] = a.gate_tab[(size_t)label * 4 * H + roff[j]];
        }
        // acc[row] += sum_k Wt[k][row] vec[k]
        auto matvec = [&](const float* __restrict__ wt, const float* vec) {
          for (int k0 = 0; k0 < H; k0 += 16) {          // H % 16 == 0 (checked at gam_create)
            float w[16][NR];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
#pragma unroll
              for (int j = 0; j < NR; ++j) w[kk][j] = wt[(size_t)(k0 + kk) * 4 * H + roff[j]];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
              const float hk = vec[k0 + kk];
#pragma unroll
              for (int j = 0; j < NR; ++j) acc[j] = fmaf(w[kk][j], hk, acc[j]);
            }
          }
        };
        auto cell = [&](const float* c_old, float* h_new, float* c_new) {   // gates (i, f, g, o) -> (h', c')
#pragma unroll
          for (int j = 0; j < NR; ++j)
            if (tid + 256 * j < 4 * H) gates[tid + 256 * j] = acc[j];
          __syncthreads();
          for (int i = tid; i < H; i += 256) {
            const float ig = gam_sigmoid_exact(gates[i]), fg = gam_sigmoid_exact(gates[H + i]);
            const float gg = tanhf(gates[2 * H + i]), og = gam_sigmoid_exact(gates[3 * H + i]);
            const float cn = fg * c_old[i] + ig * gg;
            c_new[i] = cn;
            h_new[i] = og * tanhf(cn);
          }
          __syncthreads();
        };
        matvec(a.whh_t, h_s);
        cell(c_s, hn_s, cn_s);
        for (int l = 1; l < L; ++l) {     // layers above the first: input = the NEW hidden state of the layer below
          float* st = xs + (size_t)(l - 1) * 4 * H;
          const float* below = l == 1 ? hn_s : xs + (size_t)(l - 2) * 4 * H + 2 * H;
#pragma unroll
          for (int j = 0; j < NR; ++j) acc[j] = a.bias_x[(size_t)(l - 1) * 4 * H + roff[j]];
          matvec(a.wih_x + (size_t)(l - 1) * H * 4 * H, below);
          matvec(a.whh_x + (size_t)(l - 1) * H * 4 * H, st);
          cell(st + H, st + 2 * H, st + 3 * H);
        }
      }
      const float* gtop = L > 1 ? xs + (size_t)(L - 2) * 4 * H + 2 * H : hn_s;   // g = the top layer's new hidden state
      {
        float acc0 = tid < JH ? a.bpred[tid] : 0.f, acc1 = tid + 256 < JH ? a.bpred[tid + 256] : 0.f;
        const int r0 = tid < JH ? tid : JH - 1, r1 = tid + 256 < JH ? tid + 256 : JH - 1;
        for (int k0 = 0; k0 < H; k0 += 32) {
          float w0[32], w1[32];
#pragma unroll
          for (int kk = 0; kk < 32; ++kk) {
            const int k = k0 + kk < H ? k0 + kk : H - 1;
            w0[kk] = a.wpred_t[(size_t)k * JH + r0];
            w1[kk] = a.wpred_t[(size_t)k * JH + r1];
          }
#pragma unroll
          for (int kk = 0; kk < 32; ++kk) {
            const float gk = k0 + kk < H ? gtop[k0 + kk] : 0.f;
            acc0 = fmaf(w0[kk], gk, acc0);
            acc1 = fmaf(w1[kk], gk, acc1);
          }
        }
        if (tid < JH) pp[tid] = acc0;
        if (tid + 256 < JH) pp[tid + 256] = acc1;
      }
      need_pred = false;
      __syncthreads();
    }

    // ---- joint of frames t .. t+W-1 with the current predictor state ----
    const int W = len - t < GAM_RNNT_WIN ? len - t : GAM_RNNT_WIN;
    {
      // all encoder-projection loads of the window in flight together (a plain strided loop keeps one
      // load outstanding per thread: 20 dependent L2 round trips per window)
      constexpr int NZ = GAM_RNNT_WIN * GAM_RNNT_MAXH / 256;
      float ze[NZ];
#pragma unroll
      for (int u = 0; u < NZ; ++u) {
        const int f = zfk[u] >> 16, k = zfk[u] & 0xffff;
        const int tt = t + (f < W ? f : W - 1);
        ze[u] = a.encp[((size_t)b * a.Tp + tt) * JH + k];
      }
#pragma unroll
      for (int u = 0; u < NZ; ++u) {
        const int f = zfk[u] >> 16, k = zfk[u] & 0xffff;
        if (tid + 256 * u < GAM_RNNT_WIN * JH) zw[f * ZLD + k] = fmaxf(ze[u] + pp[k], 0.f);
      }
    }
    __syncthreads();
    // logits[f][v] = bout[v] + sum_k z[f][k] * wout[v][k]: one 16x16 MFMA tile per 16 classes
    for (int nt = wave; nt * 16 < VP; nt += 4) {
      int v = nt * 16 + li;
      const int vc = v < V ? v : V - 1;
      // W_out rows: from the LDS copy when the whole matrix fits (char vocabularies), else L2 -- two instantiations of the
      // loop with address-space-qualified loads, never one generic pointer (which makes every load a FLAT instruction)
      const float* zr = zw + li * ZLD + 4 * lg4;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      auto tile = [&](auto load4, const float* wr) {
        for (int k0 = 0; k0 + 64 <= JH; k0 += 64) {      // 4 loads in flight per step
          f32x4 wf[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) wf[u] = load4(wr + k0 + 16 * u);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const f32x4 zf = gam_rc_lds4(zr + k0 + 16 * u);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zf.x, wf[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zf.y, wf[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zf.z, wf[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zf.w, wf[u].w, acc, 0, 0, 0);
          }
        }
        for (int k0 = JH / 64 * 64; k0 + 16 <= JH; k0 += 16) {
          const f32x4 zf = gam_rc_lds4(zr + k0);
          const f32x4 wf = load4(wr + k0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zf.x, wf.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zf.y, wf.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zf.z, wf.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zf.w, wf.w, acc, 0, 0, 0);
        }
      };
      if (wout_l != nullptr) tile([](const float* p) { return gam_rc_lds4(p); }, wout_l + (size_t)vc * WLD + 4 * lg4);
      else tile([](const float* p) { return gam_rc_glb4(p); }, a.wout + (size_t)vc * JH + 4 * lg4);
      // C/D: col = lane&15 = class, row = 4*(lane>>4) + r = frame
      if (v < V) {
        const float bo = a.bout[v];
#pragma unroll
        for (int r = 0; r < 4; ++r) lgw[(4 * lg4 + r) * LLD + v] = acc[r] + bo;
      }
    }
    __syncthreads();
    // per-frame argmax (first max) + log-sum-exp: wave w takes frames w, w+4, ...
    for (int f = wave; f < W; f += 4) {
      const float* lr = lgw + f * LLD;
      float best = -INFINITY;
      int bi = 0x7fffffff;
      for (int v = lane; v < V; v += 64) {
        const float x = lr[v];
        if (x > best || (x == best && v < bi)) { best = x; bi = v; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      float se = 0.f;
      if (a.dump != nullptr) {
        for (int v = lane; v < V; v += 64) se += expf(lr[v] - best);
        se = gam_wave_sum(se);
      }
      if (lane == 0) { lab_s[f] = bi; mx_s[f] = best; lse_s[f] = best + logf(se); }
    }
    __syncthreads();
    // first non-blank frame of the window (uniform scan, W <= 16)
    int fstar = W;
    for (int f = 0; f < W; ++f)
      if (lab_s[f] != blank) { fstar = f; break; }
    const int n_eval = fstar < W ? fstar + 1 : W;   // joint evaluations the sequential loop performs
    if (a.dump != nullptr) {
      for (int f = 0; f < n_eval; ++f) {
        if (n_dump + f < a.dump_cap) {
          float* dp = a.dump + ((size_t)b * a.dump_cap + n_dump + f) * V;
          const float lse = lse_s[f];
          for (int v = tid; v < V; v += 256) dp[v] = lgw[f * LLD + v] - lse;
        }
      }
    }
    n_dump += n_eval;
    if (fstar == W) {            // W blank frames
      t += W;
      sym = 0;
    } else {                     // emission at frame t + fstar (decoding.py:175-178)
      const int k = lab_s[fstar];
      const int te = t + fstar;
      if (fstar > 0) sym = 0;
      if (tid == 0 && n_out < a.cap) {
        a.ids[(size_t)b * a.cap + n_out] = k;
        a.frames[(size_t)b * a.cap + n_out] = te;
      }
      ++n_out;
      ++sym;
      label = k;
      for (int i = tid; i < H; i += 256) { h_s[i] = hn_s[i]; c_s[i] = cn_s[i]; }
      for (int i = tid; i < (L - 1) * 2 * H; i += 256) {   // commit (h', c') of the layers above the first
        const int l1 = i / (2 * H), r = i - l1 * 2 * H;
        xs[(size_t)l1 * 4 * H + r] = xs[(size_t)l1 * 4 * H + 2 * H + r];
      }
      need_pred = true;
      if (sym >= a.max_symbols) { t = te + 1; sym = 0; }   // frame advances regardless (decoding.py:189-205)
      else t = te;
    }
    __syncthreads();
  }
  if (tid == 0) {
    a.counts[b] = n_out < a.cap ? n_out : a.cap;
    if (a.dump_count != nullptr) a.dump_count[b] = n_dump;
  }
}

// ---------------------------------------------------------------------------------
// Emotion head (reference gigaam/model.py:272-285, GigaAMEmo.get_probs): mean of the encoder
// output over time -> Linear(d_model, n_classes) -> softmax.  One workgroup per utterance;
// encoded is channel-first [B, D, Tp], so a wave sums one channel's contiguous Tp frames.
// lens == nullptr: all Tp frames (what the reference's avg_pool1d over the whole axis does for
// its single, unpadded file); otherwise the first lens[b] frames.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gam_emo_head_kernel(const float* __restrict__ enc, const int* __restrict__ lens,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ probs, int D, int Tp, int NC) {
  extern __shared__ float gam_smem_emo[];   // [D] pooled + [NC] logits
  float* pooled = gam_smem_emo;
  float* logit = gam_smem_emo + D;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int n = lens != nullptr ? lens[b] : Tp;
  n = n < 1 ? 1 : (n > Tp ? Tp : n);
  for (int c = wave; c < D; c += 4) {
    const float* r = enc + ((size_t)b * D + c) * Tp;
    float s = 0.f;
    for (int t = lane; t < n; t += 64) s += r[t];
    s = gam_wave_sum(s);
    if (lane == 0) pooled[c] = s / (float)n;
  }
  __syncthreads();
  for (int k = wave; k < NC; k += 4) {
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s = fmaf(w[(size_t)k * D + c], pooled[c], s);
    s = gam_wave_sum(s);
    if (lane == 0) logit[k] = s + bias[k];
  }
  __syncthreads();
  if (wave == 0) {
    float mx = -INFINITY;
    for (int k = lane; k < NC; k += 64) mx = fmaxf(mx, logit[k]);
    mx = gam_wave_max(mx);
    float se = 0.f;
    for (int k = lane; k < NC; k += 64) se += expf(logit[k] - mx);
    se = gam_wave_sum(se);
    for (int k = lane; k < NC; k += 64) probs[(size_t)b * NC + k] = expf(logit[k] - mx) / se;
  }
}

// ------------------------------------------------------------------ RNN-T per-step entry points (r04)
// The reference exposes its predictor and joint as sub-modules (gigaam/decoder.py:41-47 RNNTJoint.joint, :85-102
// RNNTDecoder.predict); the greedy loop of this path never calls them (gam_rnnt_greedy evaluates both inside one launch per
// batch), but a caller that takes the head apart -- the reference's ONNX export, a custom beam search -- needs them.  Plain
// fp32 kernels on the layouts gam_finalize already builds for the decode kernels; latency is not a concern here.

// One LSTM step of the predictor for B samples: one workgroup per sample.
//   label[b] in [0, V): embedding row; label[b] < 0 or == V: the zero input of predict(None, ...).
//   h_in / c_in [L, B, PH] (null: zero state); g_out [B, PH] = top layer's h'; h_out / c_out [L, B, PH].
struct GamPredictArgs {
  const int* label;
  const float *h_in, *c_in;
  float *g_out, *h_out, *c_out;
  const float *gate_tab, *whh_t, *wih_x, *whh_x, *bias_x;
  int B, PH, V, L;
};

__global__ __launch_bounds__(256) void gam_rnnt_predict_kernel(GamPredictArgs a) {
  extern __shared__ float gam_smem_pred[];   // x [PH] | h [PH] | gates [4 PH]
  float* xs = gam_smem_pred;
  float* hs = xs + a.PH;
  float* gs = hs + a.PH;
  const int b = blockIdx.x, tid = threadIdx.x, G = 4 * a.PH;
  int lab = a.label != nullptr ? a.label[b] : -1;
  if (lab < 0 || lab > a.V) lab = a.V;       // row V of gate_tab = W_ih . 0 + b_ih + b_hh
  for (int l = 0; l < a.L; ++l) {
    const size_t so = ((size_t)l * a.B + b) * a.PH;
    for (int k = tid; k < a.PH; k += 256) hs[k] = a.h_in != nullptr ? a.h_in[so + k] : 0.f;
    __syncthreads();
    for (int r = tid; r < G; r += 256) {
      float acc;
      const float* wh;
      if (l == 0) {
        acc = a.gate_tab[(size_t)lab * G + r];
        wh = a.whh_t;
      } else {
        acc = a.bias_x[(size_t)(l - 1) * G + r];
        const float* wi = a.wih_x + (size_t)(l - 1) * a.PH * G;
        for (int k = 0; k < a.PH; ++k) acc = fmaf(wi[(size_t)k * G + r], xs[k], acc);
        wh = a.whh_x + (size_t)(l - 1) * a.PH * G;
      }
      for (int k = 0; k < a.PH; ++k) acc = fmaf(wh[(size_t)k * G + r], hs[k], acc);
      gs[r] = acc;
    }
    __syncthreads();
    for (int k = tid; k < a.PH; k += 256) {   // gate order i, f, g, o (nn.LSTM)
      const float ig = gam_sigmoid_exact(gs[k]), fg = gam_sigmoid_exact(gs[a.PH + k]);
      const float gg = tanhf(gs[2 * a.PH + k]), og = gam_sigmoid_exact(gs[3 * a.PH + k]);
      const float c0 = a.c_in != nullptr ? a.c_in[so + k] : 0.f;
      const float c2 = fg * c0 + ig * gg;
      const float h2 = og * tanhf(c2);
      a.c_out[so + k] = c2;
      a.h_out[so + k] = h2;
      xs[k] = h2;                              // input of the next layer
      if (l == a.L - 1) a.g_out[(size_t)b * a.PH + k] = h2;
    }
    __syncthreads();
  }
}

// z[(b, t, u), :] = relu(encp[b, t, :] + predp[b, u, :])  -- the joint's hidden layer before its output projection
__global__ __launch_bounds__(256) void gam_joint_hidden_kernel(const float* encp, const float* predp, float* z, int B, int T, int U, int JH) {
  const size_t n = (size_t)B * T * U * JH;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int j = (int)(i % JH);
    const size_t row = i / JH;
    const int u = (int)(row % U);
    const size_t bt = row / U;
    const size_t bb = bt / T;
    z[i] = fmaxf(encp[bt * JH + j] + predp[(bb * U + u) * JH + j], 0.f);
  }
}
