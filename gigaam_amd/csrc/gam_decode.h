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
// one workgroup per utterance.  Phase 1: one wave per frame, argmax with torch's
// first-max tie rule -> labels in LDS.  Phase 2: keep = label != blank && (t == 0 ||
// label != label[t-1]) && t < len; ballot/popcount prefix compaction.
__global__ __launch_bounds__(256) void gam_ctc_greedy_kernel(const float* logits, const int* enc_len, int Tp, int V,
                                                             int* ids, int* frames, int* counts) {
  extern __shared__ int gam_smem_ctc[];   // labels[Tp] + wave totals[4] + base[1]
  int* lab = gam_smem_ctc;
  int* wtot = gam_smem_ctc + Tp;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int blank = V - 1;
  int len = enc_len[b];
  len = len < 0 ? 0 : (len > Tp ? Tp : len);   // decoding.py:76 clamp
  for (int t = wave; t < Tp; t += 4) {
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
  __syncthreads();
  int base = 0;
  for (int t0 = 0; t0 < Tp; t0 += 256) {
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
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wtot[w];
    const int tot = wtot[0] + wtot[1] + wtot[2] + wtot[3];
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
// One persistent workgroup per utterance runs the whole frame/symbol loop on device.
// The reference re-runs predict(last_label, state) on every step although its inputs
// only change after a non-blank emission (decoding.py:156-160,175-178); here the
// candidate (g, h', c') and W_pred.g are computed once per commit and reused.  The
// encoder half of the joint (W_enc.f_t + b) is hoisted to one GEMM over all frames, and
// W_ih.embed[v] + b_ih + b_hh is a [V, 4H] table built at finalize.
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
};

#define GAM_RNNT_MAXH 512
#define GAM_RNNT_RPT 8   // gate rows per thread: 4*H <= 2048
#define GAM_RNNT_MAXV 2048

__global__ __launch_bounds__(256) void gam_rnnt_greedy_kernel(GamRnntArgs a) {
  __shared__ float h_s[GAM_RNNT_MAXH], c_s[GAM_RNNT_MAXH];      // committed state
  __shared__ float hn_s[GAM_RNNT_MAXH], cn_s[GAM_RNNT_MAXH];    // candidate state (= g)
  __shared__ float gates[4 * GAM_RNNT_MAXH];
  __shared__ float pp[GAM_RNNT_MAXH];                           // W_pred.g + b_pred
  __shared__ float zj[GAM_RNNT_MAXH];                           // relu(enc + pred)
  __shared__ float lg[GAM_RNNT_MAXV];
  __shared__ float red_v[4];
  __shared__ int red_i[4];
  __shared__ float lse_s;
  __shared__ int k_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int H = a.H, JH = a.JH, V = a.V, blank = a.V - 1;
  int len = a.enc_len[b];
  len = len < 0 ? 0 : (len > a.Tp ? a.Tp : len);

  for (int i = tid; i < H; i += 256) { h_s[i] = 0.f; c_s[i] = 0.f; }
  __syncthreads();

  int label = V;       // gate_tab row V: zero embedding (predict(None, None), decoder.py:97-100)
  int n_out = 0, n_dump = 0;
  bool need_pred = true;

  for (int t = 0; t < len; ++t) {
    const float* ef = a.encp + ((size_t)b * a.Tp + t) * JH;
    for (int sym = 0; sym < a.max_symbols; ++sym) {
      if (need_pred) {
        // ---- LSTM cell: gates = tab[label] + W_hh.h  (gate order i,f,g,o) ----
        // thread = gate row(s), W_hh^T rows are contiguous over the gate index, so a wave
        // reads 256 B per k; up to GAM_RNNT_RPT rows per thread give independent loads.
        {
          float acc[GAM_RNNT_RPT];
#pragma unroll
          for (int j = 0; j < GAM_RNNT_RPT; ++j) {
            const int r = tid + 256 * j;
            acc[j] = r < 4 * H ? a.gate_tab[(size_t)label * 4 * H + r] : 0.f;
          }
          for (int k = 0; k < H; ++k) {
            const float hk = h_s[k];
            const float* wt = a.whh_t + (size_t)k * 4 * H + tid;
#pragma unroll
            for (int j = 0; j < GAM_RNNT_RPT; ++j)
              if (tid + 256 * j < 4 * H) acc[j] = fmaf(wt[256 * j], hk, acc[j]);
          }
#pragma unroll
          for (int j = 0; j < GAM_RNNT_RPT; ++j)
            if (tid + 256 * j < 4 * H) gates[tid + 256 * j] = acc[j];
        }
        __syncthreads();
        for (int i = tid; i < H; i += 256) {
          const float ig = gam_sigmoid(gates[i]), fg = gam_sigmoid(gates[H + i]);
          const float gg = tanhf(gates[2 * H + i]), og = gam_sigmoid(gates[3 * H + i]);
          const float cn = fg * c_s[i] + ig * gg;
          cn_s[i] = cn;
          hn_s[i] = og * tanhf(cn);
        }
        __syncthreads();
        {
          float acc0 = tid < JH ? a.bpred[tid] : 0.f, acc1 = tid + 256 < JH ? a.bpred[tid + 256] : 0.f;
#pragma unroll 4
          for (int k = 0; k < H; ++k) {
            const float gk = hn_s[k];
            const float* wt = a.wpred_t + (size_t)k * JH + tid;
            if (tid < JH) acc0 = fmaf(wt[0], gk, acc0);
            if (tid + 256 < JH) acc1 = fmaf(wt[256], gk, acc1);
          }
          if (tid < JH) pp[tid] = acc0;
          if (tid + 256 < JH) pp[tid + 256] = acc1;
        }
        need_pred = false;
        __syncthreads();
      }
      // ---- joint: logits = W_out . relu(enc_t + pred) + b_out ----
      for (int i = tid; i < JH; i += 256) zj[i] = fmaxf(ef[i] + pp[i], 0.f);
      __syncthreads();
      {
        // one wave per class: lanes stride over the joint dimension (coalesced 256 B
        // reads of W_out rows), z is hoisted into registers, wave-reduce per class.
        float zr[GAM_RNNT_MAXH / 64];
#pragma unroll
        for (int j = 0; j < GAM_RNNT_MAXH / 64; ++j) zr[j] = lane + 64 * j < JH ? zj[lane + 64 * j] : 0.f;
        for (int v = wave; v < V; v += 4) {
          const float* wr = a.wout + (size_t)v * JH + lane;
          float acc = 0.f;
#pragma unroll
          for (int j = 0; j < GAM_RNNT_MAXH / 64; ++j)
            if (lane + 64 * j < JH) acc = fmaf(wr[64 * j], zr[j], acc);
          acc = gam_wave_sum(acc);
          if (lane == 0) lg[v] = acc + a.bout[v];
        }
      }
      __syncthreads();
      // ---- argmax (first max) ----
      float best = -INFINITY;
      int bi = 0x7fffffff;
      for (int v = tid; v < V; v += 256) {
        const float x = lg[v];
        if (x > best || (x == best && v < bi)) { best = x; bi = v; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      if (lane == 0) { red_v[wave] = best; red_i[wave] = bi; }
      __syncthreads();
      if (tid == 0) {
        float bb = red_v[0]; int ii = red_i[0];
        for (int w = 1; w < 4; ++w)
          if (red_v[w] > bb || (red_v[w] == bb && red_i[w] < ii)) { bb = red_v[w]; ii = red_i[w]; }
        k_s = ii;
        red_v[0] = bb;
      }
      __syncthreads();
      const int k = k_s;
      if (a.dump != nullptr && n_dump < a.dump_cap) {   // log_softmax of this joint call
        const float mx = red_v[0];
        float s = 0.f;
        for (int v = tid; v < V; v += 256) s += expf(lg[v] - mx);
        s = gam_wave_sum(s);
        __syncthreads();
        if (lane == 0) red_v[wave] = s;
        __syncthreads();
        if (tid == 0) lse_s = mx + logf(red_v[0] + red_v[1] + red_v[2] + red_v[3]);
        __syncthreads();
        float* dp = a.dump + ((size_t)b * a.dump_cap + n_dump) * V;
        for (int v = tid; v < V; v += 256) dp[v] = lg[v] - lse_s;
      }
      ++n_dump;
      __syncthreads();
      if (k == blank) break;
      // ---- emit + commit (decoding.py:175-178) ----
      if (tid == 0 && n_out < a.cap) {
        a.ids[(size_t)b * a.cap + n_out] = k;
        a.frames[(size_t)b * a.cap + n_out] = t;
      }
      ++n_out;
      label = k;
      for (int i = tid; i < H; i += 256) { h_s[i] = hn_s[i]; c_s[i] = cn_s[i]; }
      need_pred = true;
      __syncthreads();
    }
  }
  if (tid == 0) {
    a.counts[b] = n_out < a.cap ? n_out : a.cap;
    if (a.dump_count != nullptr) a.dump_count[b] = n_dump;
  }
}
