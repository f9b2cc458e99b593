// gam_api.hip -- host side of libgigaam_hip.so: handle, weight re-layout, workspace,
// per-batch orchestration of the gfx950 kernels, and the extern "C" boundary declared in
// include/gigaam_hip.h.  One call per batch; nothing here synchronises the host except
// workspace growth.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/gigaam_hip.h"
#include "gam_attn.h"
#include "gam_attn16.h"
#include "gam_comm.h"
#include "gam_common.h"
#include "gam_convmod.h"
#include "gam_decode.h"
#include "gam_decode_cluster.h"
#include "gam_frontend.h"
#include "gam_gemm.h"
#include "gam_gemm16.h"
#include "gam_gemm_sp.h"
#include "gam_norm.h"
#include "gam_pack.h"
#include "gam_stem.h"

namespace {

struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
  int64_t numel() const {
    int64_t n = 1;
    for (auto s : shape) n *= s;
    return n;
  }
};

struct DevBuf {
  float* p = nullptr;
  size_t cap = 0;  // floats
};

struct W16 {            // split-fp16 planes of one weight matrix (gam_gemm16.h)
  _Float16* hi = nullptr;
  _Float16* lo = nullptr;
  _Float16* sp = nullptr;   // the same planes in the sp32 layout (gam_gemm_sp.h)
  _Float16* h16 = nullptr;  // GAM_GEMM_F16: the hi plane as the one-term kernel walks it (= hi; a tap-permuted copy for the stem conv)
  float inv = 1.0f;
};

struct LayerW {
  float *ln_ff1_w, *ln_ff1_b, *ff1_w1, *ff1_b1, *ff1_w2, *ff1_b2;
  float *ln_att_w, *ln_att_b, *wqkv, *bqkv, *wo, *bo;   // rows [W_q; W_k; W_v]
  float *wpos, *pos_u, *pos_v;  // rel_pos only
  float *ln_conv_w, *ln_conv_b, *pw1_w, *pw1_b, *dw_w, *dw_b, *cn_scale, *cn_shift, *pw2_w, *pw2_b;
  float *ln_ff2_w, *ln_ff2_b, *ff2_w1, *ff2_b1, *ff2_w2, *ff2_b2;
  float *ln_out_w, *ln_out_b;
  W16 s_ff1_w1, s_ff1_w2, s_wqkv, s_wo, s_wpos, s_pw1, s_pw2, s_ff2_w1, s_ff2_w2;
};

struct ProfEvent {
  hipEvent_t a, b;
  int cls;
};

}  // namespace

struct gam_handle {
  gam_config cfg;
  int device = 0;
  bool finalized = false;
  bool has_encoder = false, has_head = false;
  std::map<std::string, HostTensor> staged;
  std::vector<void*> owned;
  std::string err;

  // frontend
  float* dft_basis = nullptr;
  int* mel_band = nullptr;   // [3 * n_mels] per mel band: first / one-past-last bin with a non-zero weight, offset into mel_wts
  float* mel_wts = nullptr;  // the non-zero stretch of every band's filter, band after band (gam_powmel_kernel keeps it in LDS)
  int mel_nw = 0;            // floats in mel_wts
  int nf = 0, kpad = 0;
  // stem
  float *c1_w = nullptr, *c1_b = nullptr, *c2_w = nullptr, *c2_b = nullptr, *lin_w = nullptr, *lin_b = nullptr;
  W16 s_c1, s_c2, s_lin;
  int gemm_mode = 1;  // GAM_GEMM_F16X3
  int f1 = 0, f2 = 0;  // conv2d: feature bins after stage 1 / 2
  // layers
  std::vector<LayerW> layers;
  float *rot_cos = nullptr, *rot_sin = nullptr;
  float* rel_pe = nullptr;  // rel_pos: sinusoid table, row r + (max_len-1) <-> relative position r
  // heads
  float *ctc_w = nullptr, *ctc_b = nullptr;
  float *emo_w = nullptr, *emo_b = nullptr;
  float *jn_enc_w = nullptr, *jn_enc_b = nullptr, *jn_pred_t = nullptr, *jn_pred_b = nullptr;
  float *jn_out_w = nullptr, *jn_out_b = nullptr, *lstm_whh_t = nullptr, *lstm_tab = nullptr;
  float *lstm_whh_q = nullptr, *jn_pred_q = nullptr;   // [k/4][row][4] re-layouts for the cluster decode kernel
  float* jn_pred_w = nullptr;   // W_pred as uploaded ([JH, PH] row-major): the GEMM form of gam_rnnt_joint
  DevBuf jz, jp, jl;            // gam_rnnt_joint workspace: hidden layer, predictor projection, logits
  float *lstm_wih_x = nullptr, *lstm_whh_x = nullptr, *lstm_bias_x = nullptr;   // predictor layers above the first (gam_decode.h)
  int use_rowscale = 1;         // GAM_ROWSCALE=0: no per-row pre-scale of the LayerNorm-produced GEMM operands (A/B switch)
  int use_range = 1;            // GAM_RANGE=0: no range guard on the unscaled operands (A/B switch)
  int ncu = 256;                // compute units of the device (hipDeviceAttributeMultiprocessorCount)
  int rnnt_cluster = -1;        // GAM_RNNT_CLUSTER: 0 = one workgroup per utterance, N = force N per utterance, -1 = auto
  int rnnt_coop = 1;            // GAM_RNNT_COOP=0: plain instead of cooperative launch of the cluster kernel
  int rnnt_force_timeout = 0;   // GAM_RNNT_FORCE_TIMEOUT=1 (test hook): odd utterances' clusters report a failed hand-off
  int rnnt_no_repair = 0;       // GAM_RNNT_NO_REPAIR=1 (test hook): no repair pass behind the cluster kernel
  int rnnt_exclusive = 0;       // GAM_RNNT_EXCLUSIVE=1: decode workgroups ask for their CU's whole LDS, so nothing that uses LDS is placed beside them
                                // (r05's protection; not needed since r06 removed the packed-fp32 instructions that a neighbour's MFMAs disturbed --
                                // tests/test_hip_hardening.py runs the decode beside another stream's GEMM both ways)
  DevBuf rnnt_x;                // hand-off granules + status word of the cluster kernel
  DevBuf rnnt_audit;            // -DGAM_RC_AUDIT=1 diagnosis builds: the cluster kernel's audit log
  bool audit_pending = false; int audit_last_c = 0; long audit_decodes = 0, audit_hit_decodes = 0;
  // The decode class owns ONE set of scratch per handle (tok / logits / encp / rnnt_x / jz / jp / jl / dec_splitk_ws).  Every decode-class
  // entry point records dec_evt behind its last launch; a decode-class call on ANOTHER stream waits for it first (DecodeScope), so two
  // decodes of one handle are ordered whatever streams the caller puts them on (an overlapped decode on the side stream followed
  // by a serial one on the launch stream overwrote the scratch the first was still reading).
  hipEvent_t dec_evt = nullptr;
  hipStream_t dec_stream = nullptr;
  bool dec_evt_set = false;

  // workspace (grow-only)
  DevBuf wavp, spec, img, c2, xin, y1, x, y, yr, hbuf, qkv, ctx, ubuf, zbuf, tok, logits, encp, pbuf, aplanes;
  int use_sp = 1;     // large-M GEMMs on the LDS-DMA sp32 kernel (GAM_SP=0 disables)
  int sp_min_m = GAM_SP_MIN_M;   // GAM_SP_MIN_M overrides (tests force the sp path at small sizes)
  DevBuf op_planes, op_sp, splitk_ws;   // gam_op_gemm operand planes; split-K partial sums
  DevBuf dec_splitk_ws;                 // split-K partial sums of the DECODE class's GEMMs: a decode may run on a side stream beside
                                        // the next batch's encoder (r05), so it shares no scratch with it (tok / logits / encp / rnnt_x
                                        // are the decode's alone already)
  int last_rows = 0, last_rows_padded = 0;   // token rows the layers of the last gam_encode* call ran on / would run on padded (gam_last_encode_rows)
  DevBuf pack_idx;      // packed rows (gam_pack.h): cu [B + 1] | row_t [rows] | row_src [rows], as ints
  int use_pack = 1;     // GAM_PACK=0: ragged batches keep the padded row layout even when the caller gave host lengths (A/B switch); 2: packed
                        // rows also below graph_max_rows token rows (tests: the small cases)
  int use_splitk = 1;   // GAM_SPLITK=0 disables split-K for small grids
  int fuse_reduce = 1;  // GAM_FUSE_REDUCE=0: the split-K reduce of a residual GEMM stays a kernel of its own (A/B switch)
  // hipGraph replay of the Conformer-layer launch sequence for small batches (launch-bound: a 5 s clip is
  // ~450 launches of 5-25 us).  Keyed on the shape; captured the second time a shape is seen (the first
  // call sizes every workspace and sets the kernel attributes); dropped whenever a workspace buffer moves.
  int use_graph = 1;            // GAM_GRAPH=0 disables
  int graph_max_rows = 2048;    // only below the large-batch (sp32) regime
  uint64_t ws_generation = 0;   // bumped by every workspace (re)allocation
  struct GraphEntry { uint64_t gen = 0; int seen = 0; hipGraphExec_t exec = nullptr; };
  std::map<std::vector<int>, GraphEntry> graphs;
  long graph_replays = 0, graph_captures_failed = 0;
  int graph_count = 0;          // instantiated graphs alive (capped at 32 shapes)
  hipStream_t cap_stream = nullptr;   // private stream for captures (the caller's may be the legacy default stream)
  int* lens = nullptr;  // 4 * maxB ints: len0, len1, len2, enc_len
  int lens_cap = 0;
  DevBuf rsbuf, op_rs;  // per-row 2^-e of the LayerNorm-produced GEMM operand currently in y / yr; gam_op_gemm's
  int* range_flag = nullptr;   // device int: set when an unscaled sp32 tensor left fp16's range (gam_range_flag)

  // profiler
  int prof_on = 0;   // 0 off, 1 every launch, 2 GEMM family only
  std::vector<ProfEvent> prof_events;
  size_t prof_used = 0;
  double prof_work[GAM_PF_NCLASS] = {0};
  double prof_bytes[GAM_PF_NCLASS] = {0};   // algorithmic operand + result bytes (GEMM classes)
  int64_t prof_launches[GAM_PF_NCLASS] = {0};
};

namespace {

int fail(gam_handle* h, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (h) h->err = buf;
  return code;
}

#define HIPCHK(h, expr)                                                                        \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return fail(h, -2, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

float* upload(gam_handle* h, const std::vector<float>& v) {
  float* d = nullptr;
  if (hipMalloc(&d, std::max<size_t>(v.size(), 4) * sizeof(float)) != hipSuccess) return nullptr;
  if (!v.empty() && hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  h->owned.push_back(d);
  return d;
}

int ensure(gam_handle* h, DevBuf& b, size_t floats) {
  if (floats <= b.cap) return 0;
  if (b.p) HIPCHK(h, hipFree(b.p));
  b.p = nullptr;
  b.cap = 0;
  HIPCHK(h, hipMalloc(&b.p, floats * sizeof(float)));
  b.cap = floats;
  ++h->ws_generation;   // captured graphs hold the old pointers
  return 0;
}

uint16_t float_to_half_bits(float f) {   // round to nearest even, subnormals kept
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x7f800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));
  if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);  // rounds to >= 65520 -> inf
  if (x < 0x38800000u) {                                      // below 2^-14: subnormal half
    if (x < 0x33000000u) return (uint16_t)sign;               // < 2^-25 -> 0
    const int e = (int)(x >> 23);
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    const int shift = 126 - e;                                // 14..24
    const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    m >>= shift;
    if (rem > half || (rem == half && (m & 1u))) ++m;
    return (uint16_t)(sign | m);
  }
  const uint32_t e = (x >> 23) - 112u;
  uint32_t m = x & 0x7fffffu;
  uint32_t h = (e << 10) | (m >> 13);
  const uint32_t rem = m & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
  return (uint16_t)(sign | h);
}

float half_to_float(uint16_t x) {
  const uint32_t sign = (x >> 15) & 1, exp = (x >> 10) & 0x1f, man = x & 0x3ff;
  uint32_t f;
  if (exp == 0) {
    if (man == 0) f = sign << 31;
    else {
      int e = -1; uint32_t m = man;
      do { ++e; m <<= 1; } while (!(m & 0x400));
      f = (sign << 31) | ((uint32_t)(127 - 15 - e) << 23) | ((m & 0x3ff) << 13);
    }
  } else if (exp == 31) f = (sign << 31) | 0x7f800000u | (man << 13);
  else f = (sign << 31) | ((exp + 127 - 15) << 23) | (man << 13);
  float r; memcpy(&r, &f, 4); return r;
}


float half_bits_to_float(uint16_t x) { return half_to_float(x); }

// W * 2^shift = hi + lo with hi, lo in fp16; the power-of-two scale puts max|W| near 2^8 so
// that lo (~2^-11 |W|) stays in the normal fp16 range for all but negligible entries.
// conv_taps > 0 (K = conv_taps * C, k = tap * C + c): the sp32 copy orders its k-tiles (32-channel block, tap) -- taps
// innermost, the order gam_gemm_sp_kernel walks an implicit-GEMM A operand in; the fp16 planes keep k as given.
int make_split(gam_handle* h, const std::vector<float>& w, W16& out, int K = 0, int conv_taps = 0) {
  float mx = 0.f;
  for (float v : w) mx = std::max(mx, fabsf(v));
  int shift = 0;
  if (mx > 0.f && std::isfinite(mx)) shift = (int)floorf(log2f(256.0f / mx));
  shift = std::max(-24, std::min(24, shift));
  const float sc = ldexpf(1.0f, shift);
  std::vector<uint16_t> hi(w.size()), lo(w.size());
  for (size_t i = 0; i < w.size(); ++i) {
    const float v = w[i] * sc;
    const uint16_t hb = float_to_half_bits(v);
    hi[i] = hb;
    lo[i] = float_to_half_bits(v - half_bits_to_float(hb));
  }
  const size_t bytes = std::max<size_t>(w.size(), 8) * 2 + 64;
  void *dh = nullptr, *dl = nullptr;
  if (hipMalloc(&dh, bytes) != hipSuccess || hipMalloc(&dl, bytes) != hipSuccess) return -2;
  h->owned.push_back(dh);
  h->owned.push_back(dl);
  if (hipMemcpy(dh, hi.data(), w.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return -2;
  if (hipMemcpy(dl, lo.data(), w.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return -2;
  out.hi = (_Float16*)dh;
  out.lo = (_Float16*)dl;
  out.h16 = out.hi;
  out.inv = ldexpf(1.0f, -shift);
  if (conv_taps > 0 && K > 0 && (K / conv_taps) % 64 == 0 && w.size() % (size_t)K == 0) {
    // one-term kernel, implicit-GEMM A operand: k-tile = 64 channels of one tap, taps innermost within a 64-channel block
    const size_t C = (size_t)K / conv_taps, cb64 = C / 64;
    std::vector<uint16_t> hp(w.size());
    for (size_t n = 0; n < w.size() / K; ++n)
      for (size_t cb = 0; cb < cb64; ++cb)
        for (size_t t = 0; t < (size_t)conv_taps; ++t)
          for (size_t j = 0; j < 64; ++j)
            hp[n * K + (cb * conv_taps + t) * 64 + j] = hi[n * K + t * C + cb * 64 + j];
    void* dp = nullptr;
    if (hipMalloc(&dp, hp.size() * 2 + 64) != hipSuccess) return -2;
    h->owned.push_back(dp);
    if (hipMemcpy(dp, hp.data(), hp.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return -2;
    out.h16 = (_Float16*)dp;
  }
  if (K > 0 && K % 32 == 0 && w.size() % (size_t)K == 0) {
    // the same planes in the sp32 layout: row n, k-block kb -> [hi x32 | lo x32] (gam_gemm_sp.h)
    std::vector<uint16_t> sp(w.size() * 2);
    for (size_t i = 0; i < w.size(); ++i) {
      const size_t n = i / K, k = i % K;
      size_t kb = k / 32;
      if (conv_taps > 0) {
        const size_t cblocks = (size_t)K / conv_taps / 32, t = kb / cblocks, cb = kb % cblocks;
        kb = cb * conv_taps + t;
      }
      const size_t o = n * 2 * (size_t)K + kb * 64 + (k % 32);
      sp[o] = hi[i];
      sp[o + 32] = lo[i];
    }
    void* ds = nullptr;
    if (hipMalloc(&ds, sp.size() * 2 + 64) != hipSuccess) return -2;
    h->owned.push_back(ds);
    if (hipMemcpy(ds, sp.data(), sp.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return -2;
    out.sp = (_Float16*)ds;
  }
  return 0;
}

const HostTensor* find(gam_handle* h, const std::string& key) {
  auto it = h->staged.find(key);
  return it == h->staged.end() ? nullptr : &it->second;
}

// ---- profiler: one event pair per launch, on the launch stream ----
struct ProfScope {
  gam_handle* h;
  hipStream_t s;
  ProfEvent* ev = nullptr;
  ProfScope(gam_handle* h_, hipStream_t s_, int cls, double work) : h(h_), s(s_) {
    if (!h->prof_on || (h->prof_on == 2 && cls != GAM_PF_GEMM && cls != GAM_PF_CONV2)) return;
    if (h->prof_used == h->prof_events.size()) {
      ProfEvent e;
      if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
      h->prof_events.push_back(e);
    }
    ev = &h->prof_events[h->prof_used++];
    ev->cls = cls;
    h->prof_work[cls] += work;
    h->prof_launches[cls] += 1;
    hipEventRecord(ev->a, s);
  }
  ~ProfScope() {
    if (ev) hipEventRecord(ev->b, s);
  }
};

// A split-K reduce left to the consumer: the LayerNorm that follows a residual GEMM sums the slices itself (gam_norm.h).
// The record carries everything the stand-alone reduce pass would need (`full` + `act`), so a consumer that does not match
// what was deferred -- another x, another shape, a GEMM launched in between -- gets the reduce kernel instead of stale sums.
struct PendingReduce {
  const float* part = nullptr;
  int nsplit = 0;
  const float* bias = nullptr;
  const float* resid = nullptr;
  float alpha = 1.0f;
  GamGemmArgs full;     // the GEMM as launched without split-K: C / M / N / ldc identify the consumer that may fuse
  int act = 0;
  void clear() { part = nullptr; nsplit = 0; }
};

// run the deferred reduce as its own pass (the fall-back of every path that cannot fuse it)
int flush_pending(gam_handle* h, hipStream_t s, PendingReduce* p) {
  if (p == nullptr || p->part == nullptr) return 0;
  const hipError_t e = gam_launch_splitk_reduce(p->full, p->act, p->nsplit, s);
  const int M = p->full.M, N = p->full.N;
  p->clear();
  if (e != hipSuccess) return fail(h, -2, "split-K reduce launch (M=%d N=%d): %s", M, N, hipGetErrorString(e));
  return 0;
}

// the split-fp16 arithmetic family: GAM_GEMM_F16X3 (three terms, fp32-equivalent, the default) and GAM_GEMM_F16 (one term, opt-in)
static inline bool split_mode(const gam_handle* h) { return h->gemm_mode != GAM_GEMM_F32; }

int gemm(gam_handle* h, hipStream_t s, const GamGemmArgs& a_in, int act, int cls = GAM_PF_GEMM, const W16* w16 = nullptr,
         PendingReduce* defer = nullptr) {
  GamGemmArgs a = a_in;
  if (int r = flush_pending(h, s, defer)) return r;   // (an unconsumed deferral must not be overwritten: its C would stay unreduced)
  a.range_flag = h->use_range ? h->range_flag : nullptr;
  if (!split_mode(h)) a.c_guard = 0;   // fp32 consumers have no range limit
  if (!split_mode(h) || w16 == nullptr || w16->hi == nullptr) a.a_rs = nullptr;   // exact-fp32 path: A is never scaled
  ProfScope ps(h, s, cls, 2.0 * (double)a.M * (double)a.N * (double)a.K);
  if (ps.ev) {   // unique bytes: A (overlapping rows counted once), W, C (+ residual)
    const double abytes = a.a_mode == 0 ? ((double)(a.M - 1) * (double)std::min<long>(a.lda, a.K) + a.K) * 4.0
                                        : (double)a.M / a.conv_f2 * 2.0 * a.conv_fp * a.conv_c * 4.0;
    h->prof_bytes[cls] += abytes * (a.n_switch > 0 ? 2.0 : 1.0) + 4.0 * a.N * a.K + 4.0 * a.M * a.N * (a.R ? 2.0 : 1.0);   // (n_switch: two A operands)
  }
  hipError_t e;
  const bool f16 = split_mode(h) && w16 != nullptr && w16->hi != nullptr;
  if (f16 && a.Asp != nullptr) {
    if (w16->sp == nullptr) return fail(h, -2, "sp32 A without sp32 W planes");
    a.Whi = w16->hi; a.Wlo = w16->lo; a.wscale_inv = w16->inv;
    a.Wsp = w16->sp;
    a.h16 = 0;
    if (a.a_fmt == 2) {
      // one fp16 MFMA per product: plain-fp16 operands (the producers wrote format 2), and the kernel is told HALF the
      // reduction length -- 64 fp16 values fill the 128-byte line that holds 32 (hi, lo) pairs in the three-term layout
      if (a.K % 64 != 0 || (a.a_mode == 0 ? a.lda % 64 != 0 : a.conv_c % 64 != 0) || w16->h16 == nullptr)
        return fail(h, -2, "GAM_GEMM_F16: K / row pitch of a GEMM operand is not a multiple of 64 (M=%d N=%d K=%d)", a.M, a.N, a.K);
      a.h16 = 1; a.K /= 2; a.lda /= 2; a.conv_c /= 2;
      a.Wsp = w16->h16;
    }
    GamSpPlan plan = gam_gemm_sp_plan(a.M, a.N, a.K, a.a_mode, h->ncu);
    if (!h->use_splitk) plan.s = 1;   // GAM_SPLITK=0: the A/B switch covers this path too (tile shape as planned)
    a.sp_mt = plan.mt; a.sp_nw = plan.nw; a.sp_ns = plan.ns;
    GamGemmArgs full = a;
    if (plan.s > 1) {   // small grid: S slices of K leave partial sums, the reduce pass applies the epilogue
      if (int r = ensure(h, h->splitk_ws, (size_t)plan.s * a.M * a.N + 64)) return r;
      a.splitk = plan.s; a.partial = h->splitk_ws.p;
    }
    e = gam_launch_gemm_sp(a, act, s);
    if (e != hipSuccess) return fail(h, -2, "sp gemm launch (M=%d N=%d K=%d): %s", a.M, a.N, a.K, hipGetErrorString(e));
    if (plan.s > 1 && defer != nullptr && h->fuse_reduce && act == GAM_ACT_NONE && !a.c_split && !a.c_guard && a.lens == nullptr &&
        !a.remap && a.ldc == a.N && a.bias != nullptr && a.R != nullptr && a.ldr == a.N) {   // (the fused row builder takes both as given)
      defer->part = a.partial; defer->nsplit = plan.s; defer->bias = a.bias; defer->resid = a.R; defer->alpha = a.alpha;
      defer->full = full; defer->full.partial = a.partial; defer->act = act;
    } else if (plan.s > 1) {
      full.partial = a.partial;
      e = gam_launch_splitk_reduce(full, act, plan.s, s);
      if (e != hipSuccess) return fail(h, -2, "split-K reduce launch (M=%d N=%d): %s", a.M, a.N, hipGetErrorString(e));
    }
    return 0;
  }
  if (a.Asp != nullptr) return fail(h, -2, "sp32 A operand outside the split-fp16 GEMM mode");
  // Split-K for grids that would leave most CUs idle (single clips, short batches: 126 tokens x 768
  // columns = 6 tiles): S slices of K write partial sums, a second pass sums them in a fixed order
  // (bit-reproducible) and applies the epilogue.
  int S = 1;
  if (h->use_splitk && a.a_mode == 0 && a.K % 32 == 0) {
    const int tiles = gam_cdiv(a.M, 128) * gam_cdiv(a.N, 128), nk = a.K / 32, ncu = 256;
    if (tiles * 2 <= ncu && nk >= 4) {
      S = std::min(16, std::min(ncu / tiles, nk / 2));
      while (S > 1 && nk % S != 0) --S;
    }
  }
  GamGemmArgs full = a;
  if (S > 1) {
    DevBuf& ws = cls == GAM_PF_DECODE ? h->dec_splitk_ws : h->splitk_ws;
    if (int r = ensure(h, ws, (size_t)S * a.M * a.N + 64)) return r;
    a.splitk = S; a.ldw = a.K; a.K = a.K / S; a.partial = ws.p;
  }
  if (f16) {
    a.Whi = w16->hi; a.Wlo = w16->lo; a.wscale_inv = w16->inv;
    e = gam_launch_gemm16(a, act, s);
  } else {
    e = gam_launch_gemm(a, act, s);
  }
  if (e != hipSuccess) return fail(h, -2, "gemm launch (M=%d N=%d K=%d): %s", a.M, a.N, a.K, hipGetErrorString(e));
  if (S > 1) {
    full.partial = a.partial;
    e = gam_launch_splitk_reduce(full, act, S, s);
    if (e != hipSuccess) return fail(h, -2, "split-K reduce launch (M=%d N=%d): %s", a.M, a.N, hipGetErrorString(e));
  }
  return 0;
}

GamGemmArgs gemm_args(const float* A, long lda, const float* W, const float* bias, float* C, long ldc, int M, int N, int K) {
  GamGemmArgs g;
  memset(&g, 0, sizeof g);
  g.A = A; g.W = W; g.bias = bias; g.C = C; g.R = nullptr;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldc = ldc; g.ldr = ldc; g.alpha = 1.0f;
  g.rpb = 1; g.fdiv = 1;
  return g;
}

int layernorm(gam_handle* h, hipStream_t s, GamLnArgs a, int mode, PendingReduce* pend = nullptr) {
  if (pend != nullptr && pend->part != nullptr) {
    // fuse only what was deferred for THIS consumer: the LayerNorm's input is the GEMM's C, same rows x columns, dense
    const GamGemmArgs& g = pend->full;
    if (a.x == g.C && a.rows == g.M && a.d == g.N && g.ldc == g.N) {
      a.part = pend->part; a.nsplit = pend->nsplit; a.pbias = pend->bias; a.presid = pend->resid; a.palpha = pend->alpha;
      a.xstore = const_cast<float*>(a.x);
      pend->clear();     // consumed: nothing may fuse (or flush) these slices again
    } else if (int r = flush_pending(h, s, pend)) {
      return r;
    }
  }
  const double bytes = (double)a.rows * a.d * 4.0 * (mode == 0 ? 2.0 : 3.0);
  ProfScope ps(h, s, GAM_PF_NORM, bytes);
  hipError_t e = gam_launch_layernorm(a, mode, s);
  if (e != hipSuccess) return fail(h, -2, "layernorm launch: %s", hipGetErrorString(e));
  return 0;
}

int64_t half_up(int64_t l) { return l <= 0 ? 0 : (l + 1) / 2; }

// Orders a decode-class call behind the previous decode-class call of the handle when the two run on different streams, and records its
// own completion for the next one (see gam_handle::dec_evt).  Same stream: no wait is enqueued, stream order does it.
struct DecodeScope {
  gam_handle* h;
  hipStream_t s;
  DecodeScope(gam_handle* h_, hipStream_t s_) : h(h_), s(s_) {
    if (h->dec_evt == nullptr && hipEventCreateWithFlags(&h->dec_evt, hipEventDisableTiming) != hipSuccess) { h->dec_evt = nullptr; (void)hipGetLastError(); }
    if (h->dec_evt != nullptr && h->dec_evt_set && h->dec_stream != s) (void)hipStreamWaitEvent(s, h->dec_evt, 0);
  }
  ~DecodeScope() {
    if (h->dec_evt != nullptr && hipEventRecord(h->dec_evt, s) == hipSuccess) { h->dec_stream = s; h->dec_evt_set = true; }
  }
};

}  // namespace

// ===================================================================================
extern "C" {

int gam_abi_version(void) { return GAM_ABI_VERSION; }

const char* gam_last_error(const gam_handle* h) { return h ? h->err.c_str() : "null handle"; }

int gam_create(const gam_config* cfg, int device_id, gam_handle** out) {
  if (!cfg || !out) return -1;
  gam_handle* h = new gam_handle();
  h->cfg = *cfg;
  h->device = device_id;
  *out = h;
  if (const char* e = getenv("GAM_SP")) h->use_sp = atoi(e);
  if (const char* e = getenv("GAM_SP_MIN_M")) h->sp_min_m = atoi(e);
  if (const char* e = getenv("GAM_SPLITK")) h->use_splitk = atoi(e);
  if (const char* e = getenv("GAM_PACK")) h->use_pack = atoi(e);
  if (const char* e = getenv("GAM_FUSE_REDUCE")) h->fuse_reduce = atoi(e);
  if (const char* e = getenv("GAM_GRAPH")) h->use_graph = atoi(e);
  if (const char* e = getenv("GAM_GRAPH_MAX_ROWS")) h->graph_max_rows = atoi(e);
  if (const char* e = getenv("GAM_RNNT_CLUSTER")) h->rnnt_cluster = std::max(-1, std::min(8, atoi(e)));   // (the range gam_set_rnnt_cluster accepts)
  if (const char* e = getenv("GAM_RNNT_EXCLUSIVE")) h->rnnt_exclusive = atoi(e);
  if (const char* e = getenv("GAM_RNNT_COOP")) h->rnnt_coop = atoi(e);
  if (const char* e = getenv("GAM_RNNT_FORCE_TIMEOUT")) h->rnnt_force_timeout = atoi(e);
  if (const char* e = getenv("GAM_RNNT_NO_REPAIR")) h->rnnt_no_repair = atoi(e);
  {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && n > 0) h->ncu = n;
    int co = 0;
    if (hipDeviceGetAttribute(&co, hipDeviceAttributeCooperativeLaunch, device_id) != hipSuccess || co == 0) h->rnnt_coop = 0;
  }
  if (const char* e = getenv("GAM_ROWSCALE")) h->use_rowscale = atoi(e);
  if (const char* e = getenv("GAM_RANGE")) h->use_range = atoi(e);
  if (const char* e = getenv("GAM_GEMM_MODE"))
    h->gemm_mode = strcmp(e, "f32") == 0 ? GAM_GEMM_F32 : (strcmp(e, "f16") == 0 ? GAM_GEMM_F16 : GAM_GEMM_F16X3);
  const gam_config& c = h->cfg;
  if (c.d_model <= 0 || c.n_heads <= 0 || c.d_model % c.n_heads != 0)
    return fail(h, -1, "bad d_model/n_heads %d/%d", c.d_model, c.n_heads);
  if (c.d_model / c.n_heads != GAM_ATT_DK) return fail(h, -1, "head dim %d unsupported (kernels are built for %d)", c.d_model / c.n_heads, GAM_ATT_DK);
  if (c.d_model % 64 != 0 || c.d_model > 1024) return fail(h, -1, "d_model %d unsupported", c.d_model);
  if (c.subsampling_factor != 4) return fail(h, -1, "subsampling_factor %d unsupported", c.subsampling_factor);
  if (c.subsampling == GAM_SUBS_CONV2D && c.subs_kernel_size != 3) return fail(h, -1, "conv2d stem needs kernel 3");
  if (c.subsampling == GAM_SUBS_CONV1D && c.subs_kernel_size != 5) return fail(h, -1, "conv1d stem needs kernel 5");
  if (c.win_length != c.n_fft) return fail(h, -1, "win_length != n_fft unsupported");
  if (c.n_mels > 64 || c.n_mels != c.feat_in) return fail(h, -1, "n_mels %d / feat_in %d unsupported", c.n_mels, c.feat_in);
  if (c.head_type == GAM_HEAD_RNNT && (c.pred_rnn_layers < 1 || c.pred_rnn_layers > 4 || c.pred_hidden > GAM_RNNT_MAXH || c.joint_hidden > GAM_RNNT_MAXH || c.num_classes > GAM_RNNT_MAXV || c.joint_hidden % 16 != 0 || c.pred_hidden % 16 != 0))
    return fail(h, -1, "RNN-T head shape unsupported");
  if (hipSetDevice(device_id) != hipSuccess) return fail(h, -2, "hipSetDevice(%d) failed", device_id);
  return 0;
}

#if GAM_RC_AUDIT
static int audit_dump(gam_handle* h);
#endif
void gam_destroy(gam_handle* h) {
  if (!h) return;
  hipSetDevice(h->device);
#if GAM_RC_AUDIT
  audit_dump(h);
  fprintf(stderr, "[gam-audit] total: %ld of %ld cluster decodes logged at least one entry\n", h->audit_hit_decodes, h->audit_decodes);
#endif
  for (void* p : h->owned) hipFree(p);
  DevBuf* bufs[] = {&h->wavp, &h->spec, &h->img, &h->c2, &h->xin, &h->y1, &h->x, &h->y, &h->yr, &h->hbuf,
                    &h->qkv, &h->ctx, &h->ubuf, &h->zbuf, &h->tok, &h->logits, &h->encp, &h->pbuf, &h->aplanes, &h->op_planes, &h->op_sp, &h->splitk_ws, &h->dec_splitk_ws, &h->rsbuf, &h->op_rs, &h->rnnt_x, &h->rnnt_audit, &h->jz, &h->jp, &h->jl, &h->pack_idx};
  for (DevBuf* b : bufs)
    if (b->p) hipFree(b->p);
  if (h->lens) hipFree(h->lens);
  if (h->range_flag) hipFree(h->range_flag);
  for (auto& e : h->prof_events) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
  for (auto& g : h->graphs)
    if (g.second.exec) hipGraphExecDestroy(g.second.exec);
  if (h->cap_stream) hipStreamDestroy(h->cap_stream);
  if (h->dec_evt) hipEventDestroy(h->dec_evt);
  if (getenv("GAM_GRAPH_DEBUG")) fprintf(stderr, "[gam] graph replays %ld, failed captures %ld\n", h->graph_replays, h->graph_captures_failed);
  delete h;
}

int gam_set_weight(gam_handle* h, const char* key, const void* host_ptr, int dtype, const int64_t* shape, int ndim) {
  if (!h || !key || !host_ptr) return -1;
  if (h->finalized) return fail(h, -1, "gam_set_weight after gam_finalize");
  HostTensor t;
  for (int i = 0; i < ndim; ++i) t.shape.push_back(shape[i]);
  const int64_t n = t.numel();
  t.data.resize((size_t)n);
  switch (dtype) {
    case GAM_DTYPE_F32: memcpy(t.data.data(), host_ptr, (size_t)n * 4); break;
    case GAM_DTYPE_F16: for (int64_t i = 0; i < n; ++i) t.data[i] = half_to_float(((const uint16_t*)host_ptr)[i]); break;
    case GAM_DTYPE_BF16: for (int64_t i = 0; i < n; ++i) { uint32_t u = (uint32_t)((const uint16_t*)host_ptr)[i] << 16; memcpy(&t.data[i], &u, 4); } break;
    case GAM_DTYPE_F64: for (int64_t i = 0; i < n; ++i) t.data[i] = (float)((const double*)host_ptr)[i]; break;
    case GAM_DTYPE_I64: for (int64_t i = 0; i < n; ++i) t.data[i] = (float)((const int64_t*)host_ptr)[i]; break;
    default: return fail(h, -1, "unknown dtype %d for %s", dtype, key);
  }
  h->staged[key] = std::move(t);
  return 0;
}

#define NEED(var, key, n)                                                                          \
  const HostTensor* var = find(h, key);                                                            \
  if (!var) return fail(h, -3, "missing weight %s", std::string(key).c_str());                     \
  if (var->numel() != (int64_t)(n)) return fail(h, -3, "weight %s has %lld elements, expected %lld", std::string(key).c_str(), (long long)var->numel(), (long long)(n));

#define UP(dst, vec)                                   \
  dst = upload(h, vec);                                \
  if (!dst) return fail(h, -2, "device upload failed (%s)", #dst);

int gam_finalize(gam_handle* h) {
  if (!h) return -1;
  if (h->finalized) return 0;
  HIPCHK(h, hipSetDevice(h->device));
  const gam_config& c = h->cfg;
  const int D = c.d_model, C = c.d_model, F = c.feat_in, H = c.n_heads, dk = D / H;

  // ---------------- frontend: window-folded DFT basis + mel filterbank ----------------
  {
    const int n = c.n_fft, nf = n / 2 + 1;
    h->nf = nf;
    h->kpad = (n + 31) / 32 * 32;
    std::vector<double> win(n);
    if (const HostTensor* w = find(h, "preprocessor.featurizer.0.spectrogram.window")) {
      if (w->numel() != n) return fail(h, -3, "window has %lld taps, expected %d", (long long)w->numel(), n);
      for (int i = 0; i < n; ++i) win[i] = w->data[i];
    } else {
      for (int i = 0; i < n; ++i) win[i] = (double)(float)(0.5 - 0.5 * cos(2.0 * M_PI * i / n));  // periodic hann
    }
    std::vector<float> basis((size_t)2 * nf * h->kpad, 0.f);
    for (int k = 0; k < nf; ++k)
      for (int i = 0; i < n; ++i) {
        const long ki = ((long)k * i) % n;  // exact angle reduction
        const double ang = 2.0 * M_PI * (double)ki / n;
        basis[(size_t)k * h->kpad + i] = (float)(win[i] * cos(ang));
        basis[(size_t)(nf + k) * h->kpad + i] = (float)(-win[i] * sin(ang));
      }
    UP(h->dft_basis, basis);
    std::vector<float> fb;
    if (const HostTensor* f = find(h, "preprocessor.featurizer.0.mel_scale.fb")) {
      if (f->numel() != (int64_t)nf * c.n_mels) return fail(h, -3, "mel fb has %lld elements, expected %d", (long long)f->numel(), nf * c.n_mels);
      fb = f->data;
    } else {
      // torchaudio melscale_fbanks(htk, norm=None, f_min 0, f_max sr/2)
      fb.assign((size_t)nf * c.n_mels, 0.f);
      const double fmax = c.sample_rate / 2.0, mmax = 2595.0 * log10(1.0 + fmax / 700.0);
      std::vector<double> fpts(c.n_mels + 2);
      for (int i = 0; i < c.n_mels + 2; ++i) fpts[i] = 700.0 * (pow(10.0, (mmax * i / (c.n_mels + 1)) / 2595.0) - 1.0);
      for (int f = 0; f < nf; ++f) {
        const double fr = (double)(c.sample_rate / 2) * f / (nf - 1);
        for (int m = 0; m < c.n_mels; ++m) {
          const double down = (fr - fpts[m]) / (fpts[m + 1] - fpts[m]), up = (fpts[m + 2] - fr) / (fpts[m + 2] - fpts[m + 1]);
          fb[(size_t)f * c.n_mels + m] = (float)std::max(0.0, std::min(down, up));
        }
      }
    }
    std::vector<int> band(3 * (size_t)c.n_mels, 0);
    std::vector<float> wts;
    for (int m = 0; m < c.n_mels; ++m) {
      int lo = nf, hi = 0;
      for (int f = 0; f < nf; ++f)
        if (fb[(size_t)f * c.n_mels + m] != 0.f) { lo = std::min(lo, f); hi = std::max(hi, f + 1); }
      if (hi <= lo) lo = hi = 0;
      band[3 * m] = lo; band[3 * m + 1] = hi; band[3 * m + 2] = (int)wts.size();
      for (int f = lo; f < hi; ++f) wts.push_back(fb[(size_t)f * c.n_mels + m]);
    }
    h->mel_nw = (int)wts.size();
    UP(h->mel_wts, wts);
    void* db = nullptr;
    if (hipMalloc(&db, band.size() * sizeof(int)) != hipSuccess) return fail(h, -2, "mel band upload failed");
    h->owned.push_back(db);
    if (hipMemcpy(db, band.data(), band.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return fail(h, -2, "mel band upload failed");
    h->mel_band = (int*)db;
  }

  // ---------------- stem ----------------
  // A handle may carry only a subset of the operators (e.g. a stand-alone
  // FeatureExtractor or ConformerEncoder module on the Python side).
  const std::string pe = "encoder.pre_encode.";
  const bool build_encoder = find(h, pe + "conv.0.weight") != nullptr;
  if (!build_encoder) {
    // nothing to do
  } else if (c.subsampling == GAM_SUBS_CONV2D) {
    h->f1 = (F + 1) / 2;
    h->f2 = (h->f1 + 1) / 2;
    NEED(w0, pe + "conv.0.weight", (int64_t)C * 9);
    NEED(b0, pe + "conv.0.bias", C);
    NEED(w2, pe + "conv.2.weight", (int64_t)C * C * 9);
    NEED(b2, pe + "conv.2.bias", C);
    NEED(wl, pe + "out.weight", (int64_t)D * C * h->f2);
    NEED(bl, pe + "out.bias", D);
    UP(h->c1_w, w0->data);
    UP(h->c1_b, b0->data);
    // [n][c][kh][kw] -> [n][kh*3+kw][c]
    std::vector<float> r((size_t)C * 9 * C);
    for (int n = 0; n < C; ++n)
      for (int ci = 0; ci < C; ++ci)
        for (int t = 0; t < 9; ++t) r[((size_t)n * 9 + t) * C + ci] = w2->data[((size_t)n * C + ci) * 9 + t];
    UP(h->c2_w, r);
    if (make_split(h, r, h->s_c2, 9 * C, C % 32 == 0 ? 9 : 0)) return fail(h, -2, "split upload failed");
    UP(h->c2_b, b2->data);
    // columns c*f2+f -> f*C+c (encoder.py:126-127 flattens channel-major)
    std::vector<float> l((size_t)D * C * h->f2);
    const int f2 = h->f2;
    for (int n = 0; n < D; ++n)
      for (int ci = 0; ci < C; ++ci)
        for (int f = 0; f < f2; ++f) l[(size_t)n * C * f2 + (size_t)f * C + ci] = wl->data[(size_t)n * C * f2 + (size_t)ci * f2 + f];
    UP(h->lin_w, l);
    if (make_split(h, l, h->s_lin, C * h->f2)) return fail(h, -2, "split upload failed");
    UP(h->lin_b, bl->data);
  } else {
    const int ks = c.subs_kernel_size;
    NEED(w0, pe + "conv.0.weight", (int64_t)C * F * ks);
    NEED(b0, pe + "conv.0.bias", C);
    NEED(w2, pe + "conv.2.weight", (int64_t)C * C * ks);
    NEED(b2, pe + "conv.2.bias", C);
    // [n][f][k] -> [n][k][f]
    std::vector<float> r0((size_t)C * ks * F), r2((size_t)C * ks * C);
    for (int n = 0; n < C; ++n)
      for (int f = 0; f < F; ++f)
        for (int k = 0; k < ks; ++k) r0[((size_t)n * ks + k) * F + f] = w0->data[((size_t)n * F + f) * ks + k];
    for (int n = 0; n < C; ++n)
      for (int ci = 0; ci < C; ++ci)
        for (int k = 0; k < ks; ++k) r2[((size_t)n * ks + k) * C + ci] = w2->data[((size_t)n * C + ci) * ks + k];
    UP(h->c1_w, r0);
    if (make_split(h, r0, h->s_c1)) return fail(h, -2, "split upload failed");
    UP(h->c1_b, b0->data);
    UP(h->c2_w, r2);
    if (make_split(h, r2, h->s_c2)) return fail(h, -2, "split upload failed");
    UP(h->c2_b, b2->data);
  }

  // ---------------- layers ----------------
  const int DFF = D * c.ff_expansion_factor, ks = c.conv_kernel_size;
  h->layers.resize(build_encoder ? c.n_layers : 0);
  for (int li = 0; li < (build_encoder ? c.n_layers : 0); ++li) {
    LayerW& L = h->layers[li];
    memset(&L, 0, sizeof L);
    const std::string p = "encoder.layers." + std::to_string(li) + ".";
#define LN_(dstw, dstb, name)                 \
  {                                           \
    NEED(w_, p + name + ".weight", D);        \
    NEED(b_, p + name + ".bias", D);          \
    UP(dstw, w_->data);                       \
    UP(dstb, b_->data);                       \
  }
#define LIN_(dstw, dstb, name, nout, nin)                 \
  {                                                       \
    NEED(w_, p + name + ".weight", (int64_t)(nout) * (nin)); \
    NEED(b_, p + name + ".bias", nout);                   \
    UP(dstw, w_->data);                                   \
    UP(dstb, b_->data);                                   \
  }
    LN_(L.ln_ff1_w, L.ln_ff1_b, "norm_feed_forward1");
#define LINS_(dstw, dstb, dsts, name, nout, nin)                                     \
  {                                                                                  \
    NEED(w_, p + name + ".weight", (int64_t)(nout) * (nin));                         \
    NEED(b_, p + name + ".bias", nout);                                              \
    UP(dstw, w_->data);                                                              \
    UP(dstb, b_->data);                                                              \
    if (make_split(h, w_->data, dsts, nin)) return fail(h, -2, "split upload failed"); \
  }
    LINS_(L.ff1_w1, L.ff1_b1, L.s_ff1_w1, "feed_forward1.linear1", DFF, D);
    LINS_(L.ff1_w2, L.ff1_b2, L.s_ff1_w2, "feed_forward1.linear2", D, DFF);
    LN_(L.ln_att_w, L.ln_att_b, "norm_self_att");
    {
      NEED(wq, p + "self_attn.linear_q.weight", (int64_t)D * D);
      NEED(bq, p + "self_attn.linear_q.bias", D);
      NEED(wk, p + "self_attn.linear_k.weight", (int64_t)D * D);
      NEED(bk, p + "self_attn.linear_k.bias", D);
      NEED(wv, p + "self_attn.linear_v.weight", (int64_t)D * D);
      NEED(bv, p + "self_attn.linear_v.bias", D);
      std::vector<float> w(wq->data), b(bq->data);
      w.insert(w.end(), wk->data.begin(), wk->data.end());
      w.insert(w.end(), wv->data.begin(), wv->data.end());
      b.insert(b.end(), bk->data.begin(), bk->data.end());
      b.insert(b.end(), bv->data.begin(), bv->data.end());
      UP(L.wqkv, w);
      UP(L.bqkv, b);
      if (make_split(h, w, L.s_wqkv, D)) return fail(h, -2, "split upload failed");
    }
    LINS_(L.wo, L.bo, L.s_wo, "self_attn.linear_out", D, D);
    if (c.self_attention_model == GAM_ATT_REL_POS) {
      NEED(wp, p + "self_attn.linear_pos.weight", (int64_t)D * D);
      NEED(pu, p + "self_attn.pos_bias_u", D);
      NEED(pv, p + "self_attn.pos_bias_v", D);
      UP(L.wpos, wp->data);
      if (make_split(h, wp->data, L.s_wpos)) return fail(h, -2, "split upload failed");
      UP(L.pos_u, pu->data);
      UP(L.pos_v, pv->data);
    }
    LN_(L.ln_conv_w, L.ln_conv_b, "norm_conv");
    LINS_(L.pw1_w, L.pw1_b, L.s_pw1, "conv.pointwise_conv1", 2 * D, D);
    LIN_(L.dw_w, L.dw_b, "conv.depthwise_conv", D, ks);
    {
      NEED(g, p + "conv.batch_norm.weight", D);
      NEED(be, p + "conv.batch_norm.bias", D);
      if (c.conv_norm_type == GAM_NORM_BATCH) {
        NEED(rm, p + "conv.batch_norm.running_mean", D);
        NEED(rv, p + "conv.batch_norm.running_var", D);
        std::vector<float> sc(D), sh(D);
        for (int i = 0; i < D; ++i) {
          const float inv = 1.0f / sqrtf(rv->data[i] + 1e-5f);
          sc[i] = g->data[i] * inv;
          sh[i] = be->data[i] - rm->data[i] * sc[i];
        }
        UP(L.cn_scale, sc);
        UP(L.cn_shift, sh);
      } else {
        UP(L.cn_scale, g->data);
        UP(L.cn_shift, be->data);
      }
    }
    LINS_(L.pw2_w, L.pw2_b, L.s_pw2, "conv.pointwise_conv2", D, D);
    LN_(L.ln_ff2_w, L.ln_ff2_b, "norm_feed_forward2");
    LINS_(L.ff2_w1, L.ff2_b1, L.s_ff2_w1, "feed_forward2.linear1", DFF, D);
    LINS_(L.ff2_w2, L.ff2_b2, L.s_ff2_w2, "feed_forward2.linear2", D, DFF);
    LN_(L.ln_out_w, L.ln_out_b, "norm_out");
#undef LN_
#undef LIN_
#undef LINS_
  }

  // ---------------- rotary table (encoder.py:342-355; base = pos_emb_max_len) ----------------
  if (!build_encoder) {
  } else if (c.self_attention_model == GAM_ATT_ROTARY) {
    const int half = dk / 2, n = c.pos_emb_max_len;
    std::vector<float> cs((size_t)n * half), sn((size_t)n * half);
    for (int i = 0; i < half; ++i) {
      const float inv_freq = 1.0f / powf((float)n, (float)(2 * i) / (float)dk);
      for (int t = 0; t < n; ++t) {
        const float fr = (float)t * inv_freq;
        cs[(size_t)t * half + i] = (float)cos((double)fr);
        sn[(size_t)t * half + i] = (float)sin((double)fr);
      }
    }
    UP(h->rot_cos, cs);
    UP(h->rot_sin, sn);
  } else {
    // RelPositionalEmbedding.create_pe (encoder.py:312-327): pe(r)[2i] = sin(r * w_i),
    // pe(r)[2i+1] = cos(r * w_i), w_i = exp(2i * -(ln 10000 / D)); rows ascending in r
    const int n = c.pos_emb_max_len;
    std::vector<float> pe((size_t)(2 * n - 1) * D);
    for (int i = 0; i < D / 2; ++i) {
      const float w = expf((float)(2 * i) * (float)(-(log(10000.0) / D)));
      for (int r = -(n - 1); r <= n - 1; ++r) {
        const float ang = (float)r * w;
        pe[(size_t)(r + n - 1) * D + 2 * i] = (float)sin((double)ang);
        pe[(size_t)(r + n - 1) * D + 2 * i + 1] = (float)cos((double)ang);
      }
    }
    UP(h->rel_pe, pe);
  }

  h->has_encoder = build_encoder;

  // ---------------- heads ----------------
  const bool build_head = find(h, "head.decoder_layers.0.weight") != nullptr || find(h, "head.joint.enc.weight") != nullptr ||
                          (c.head_type == GAM_HEAD_EMO && find(h, "head.weight") != nullptr);
  h->has_head = build_head;
  if (!build_head) {
  } else if (c.head_type == GAM_HEAD_CTC) {
    NEED(w, "head.decoder_layers.0.weight", (int64_t)c.num_classes * D);
    NEED(b, "head.decoder_layers.0.bias", c.num_classes);
    UP(h->ctc_w, w->data);
    UP(h->ctc_b, b->data);
  } else if (c.head_type == GAM_HEAD_EMO) {
    NEED(w, "head.weight", (int64_t)c.num_classes * D);
    NEED(b, "head.bias", c.num_classes);
    UP(h->emo_w, w->data);
    UP(h->emo_b, b->data);
  } else if (c.head_type == GAM_HEAD_RNNT) {
    const int V = c.num_classes, PH = c.pred_hidden, JH = c.joint_hidden;
    NEED(emb, "head.decoder.embed.weight", (int64_t)V * PH);
    NEED(wih, "head.decoder.lstm.weight_ih_l0", (int64_t)4 * PH * PH);
    NEED(whh, "head.decoder.lstm.weight_hh_l0", (int64_t)4 * PH * PH);
    NEED(bih, "head.decoder.lstm.bias_ih_l0", 4 * PH);
    NEED(bhh, "head.decoder.lstm.bias_hh_l0", 4 * PH);
    NEED(wp, "head.joint.pred.weight", (int64_t)JH * PH);
    NEED(bp, "head.joint.pred.bias", JH);
    NEED(we, "head.joint.enc.weight", (int64_t)JH * D);
    NEED(bee, "head.joint.enc.bias", JH);
    NEED(wo, "head.joint.joint_net.1.weight", (int64_t)V * JH);
    NEED(bo, "head.joint.joint_net.1.bias", V);
    std::vector<float> tab((size_t)(V + 1) * 4 * PH), whh_t((size_t)PH * 4 * PH), wp_t((size_t)PH * JH);
    for (int v = 0; v <= V; ++v)
      for (int r = 0; r < 4 * PH; ++r) {
        float acc = 0.f;
        if (v < V) {
          const float* wr = &wih->data[(size_t)r * PH];
          const float* e = &emb->data[(size_t)v * PH];
          for (int k = 0; k < PH; ++k) acc = fmaf(wr[k], e[k], acc);
        }
        tab[(size_t)v * 4 * PH + r] = (acc + bih->data[r]) + bhh->data[r];
      }
    for (int r = 0; r < 4 * PH; ++r)
      for (int k = 0; k < PH; ++k) whh_t[(size_t)k * 4 * PH + r] = whh->data[(size_t)r * PH + k];
    for (int r = 0; r < JH; ++r)
      for (int k = 0; k < PH; ++k) wp_t[(size_t)k * JH + r] = wp->data[(size_t)r * PH + k];
    std::vector<float> whh_q((size_t)PH * 4 * PH), wp_q((size_t)PH * JH);
    for (int r = 0; r < 4 * PH; ++r)
      for (int k = 0; k < PH; ++k) whh_q[((size_t)(k / 4) * 4 * PH + r) * 4 + (k & 3)] = whh->data[(size_t)r * PH + k];
    for (int r = 0; r < JH; ++r)
      for (int k = 0; k < PH; ++k) wp_q[((size_t)(k / 4) * JH + r) * 4 + (k & 3)] = wp->data[(size_t)r * PH + k];
    if (c.pred_rnn_layers > 1) {   // nn.LSTM layers 1 .. L-1 (decoder.py:78-83): transposed like W_hh, biases summed
      const int LX = c.pred_rnn_layers - 1;
      std::vector<float> wih_x((size_t)LX * PH * 4 * PH), whh_x((size_t)LX * PH * 4 * PH), bias_x((size_t)LX * 4 * PH);
      for (int l = 1; l <= LX; ++l) {
        const std::string sfx = "_l" + std::to_string(l);
        NEED(wi, "head.decoder.lstm.weight_ih" + sfx, (int64_t)4 * PH * PH);
        NEED(wh, "head.decoder.lstm.weight_hh" + sfx, (int64_t)4 * PH * PH);
        NEED(bi, "head.decoder.lstm.bias_ih" + sfx, 4 * PH);
        NEED(bh, "head.decoder.lstm.bias_hh" + sfx, 4 * PH);
        const size_t o = (size_t)(l - 1) * PH * 4 * PH;
        for (int r = 0; r < 4 * PH; ++r) {
          for (int k = 0; k < PH; ++k) {
            wih_x[o + (size_t)k * 4 * PH + r] = wi->data[(size_t)r * PH + k];
            whh_x[o + (size_t)k * 4 * PH + r] = wh->data[(size_t)r * PH + k];
          }
          bias_x[(size_t)(l - 1) * 4 * PH + r] = bi->data[r] + bh->data[r];
        }
      }
      UP(h->lstm_wih_x, wih_x);
      UP(h->lstm_whh_x, whh_x);
      UP(h->lstm_bias_x, bias_x);
    }
    UP(h->lstm_whh_q, whh_q);
    UP(h->jn_pred_q, wp_q);
    UP(h->lstm_tab, tab);
    UP(h->lstm_whh_t, whh_t);
    UP(h->jn_pred_t, wp_t);
    UP(h->jn_pred_b, bp->data);
    UP(h->jn_pred_w, wp->data);
    UP(h->jn_enc_w, we->data);
    UP(h->jn_enc_b, bee->data);
    UP(h->jn_out_w, wo->data);
    UP(h->jn_out_b, bo->data);
  }
  HIPCHK(h, hipMalloc(&h->range_flag, 64));
  HIPCHK(h, hipMemset(h->range_flag, 0, 64));
  h->staged.clear();
  h->finalized = true;
  return 0;
}

int64_t gam_feat_frames(const gam_handle* h, int64_t n) {
  const gam_config& c = h->cfg;
  if (c.center) return n / c.hop_length + 1;
  return n < c.win_length ? 0 : (n - c.win_length) / c.hop_length + 1;
}

int64_t gam_enc_frames(const gam_handle* h, int64_t t) {
  (void)h;
  return half_up(half_up(t));
}

// -----------------------------------------------------------------------------------
int gam_frontend(gam_handle* h, const float* wav, const int64_t* wav_len, int B, int64_t L, float* feat,
                 int64_t* feat_len, void* stream) {
  if (!h || !h->finalized) return fail(h, -1, "gam_frontend before gam_finalize");
  if (B <= 0 || L <= 0) return fail(h, -1, "bad shape B=%d L=%lld", B, (long long)L);
  const gam_config& c = h->cfg;
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(h, hipSetDevice(h->device));
  const int hop = c.hop_length, n = c.n_fft, half = n / 2;
  if (c.center && L <= half) return fail(h, -1, "input of %lld samples is too short for reflect padding (%d)", (long long)L, half);
  const int64_t Tf = gam_feat_frames(h, L);
  if (Tf <= 0) return fail(h, -1, "input of %lld samples yields no frames", (long long)L);
  // per-utterance row stride Tfa: row m = b*Tfa + t starts at sample m*hop of the padded buffer
  const int64_t need = (Tf - 1) * hop + h->kpad;
  const int64_t Tfa = (need + hop - 1) / hop;
  const int64_t Lp = Tfa * hop;
  const int lds = (2 * h->nf + 3) / 4 * 4;
  if (int r = ensure(h, h->wavp, (size_t)B * Lp + h->kpad + 64)) return r;
  if (int r = ensure(h, h->spec, (size_t)B * Tfa * lds)) return r;
  {
    ProfScope ps(h, s, GAM_PF_FRONTEND, (double)B * (L + Lp) * 4.0);
    // one extra "utterance row" of zeros is the slack the last rows' K padding reads
    dim3 grid(gam_cdiv(Lp, 256), B);
    hipLaunchKernelGGL(gam_pad_wav_kernel, grid, dim3(256), 0, s, wav, h->wavp.p, B, (long)L, (long)Lp, half, c.center);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemsetAsync(h->wavp.p + (size_t)B * Lp, 0, (h->kpad + 64) * sizeof(float), s));
  }
  GamGemmArgs g = gemm_args(h->wavp.p, hop, h->dft_basis, nullptr, h->spec.p, lds, (int)(B * Tfa), 2 * h->nf, h->kpad);
  // Exact-fp32 MFMA in every mode (like the head GEMMs).  The split-fp16 path assumes O(1) activations: the
  // samples of a quiet frame (|x| ~ 1e-3) leave a_lo in the fp16 subnormal range, the frame's spectrum keeps
  // ~14 bits and its log-mel was off by up to 2e-3 only 50 dB below the frame's own peak (tools/fe_dbg.py;
  // fp32: 1e-4 at 60 dB, the level of an fp32 matmul DFT on the CPU).  21 GFLOP: +70 us per 32 x 20 s batch.
  if (int r = gemm(h, s, g, GAM_ACT_NONE, GAM_PF_FRONTEND, nullptr)) return r;
  {
    GamPowMelArgs a;
    a.spec = h->spec.p; a.wts = h->mel_wts; a.nw = h->mel_nw; a.band = h->mel_band; a.feat = feat; a.wav_len = (const long long*)wav_len; a.feat_len = (long long*)feat_len;
    a.B = B; a.Tfa = (int)Tfa; a.Tf = (int)Tf; a.nf = h->nf; a.n_mels = c.n_mels; a.lds = lds;
    a.hop = hop; a.win = c.win_length; a.center = c.center;
    ProfScope ps(h, s, GAM_PF_FRONTEND, (double)B * Tf * (2.0 * h->nf + c.n_mels) * 4.0);
    dim3 grid(gam_cdiv(Tf, 64), B);
    const size_t pm_lds = (size_t)(64 * (h->nf + 1) + h->mel_nw) * sizeof(float);
    if (pm_lds > 160 * 1024) return fail(h, -1, "n_fft %d: the power / mel kernel's LDS image (%zu bytes) does not fit", c.n_fft, pm_lds);
    if (pm_lds > 64 * 1024) {   // (n_fft > 500; the published frontends need 53 KB)
      static std::atomic<unsigned long long> attr_devs{0};
      HIPCHK(h, gam_set_max_lds(reinterpret_cast<const void*>(gam_powmel_kernel), (int)pm_lds, attr_devs));
    }
    hipLaunchKernelGGL(gam_powmel_kernel, grid, dim3(256), pm_lds, s, a);
    HIPCHK(h, hipGetLastError());
  }
  return 0;
}

// -----------------------------------------------------------------------------------
static int encode_impl(gam_handle* h, const float* feat, const int64_t* feat_len, int B, int64_t T, float* encoded,
                       int32_t* enc_len, int n_layers_run, float* tokens_out, hipStream_t s, const int64_t* feat_len_host = nullptr) {
  if (!h || !h->finalized) return fail(h, -1, "gam_encode before gam_finalize");
  if (!h->has_encoder) return fail(h, -1, "this handle was finalized without encoder weights");
  const gam_config& c = h->cfg;
  if (B <= 0 || T <= 0) return fail(h, -1, "bad shape B=%d T=%lld", B, (long long)T);
  HIPCHK(h, hipSetDevice(h->device));
  const int D = c.d_model, C = D, F = c.feat_in, H = c.n_heads, dk = D / H, DFF = D * c.ff_expansion_factor;
  const int T1 = (int)half_up(T), T2 = (int)half_up(T1), Tv = T2;
  if (Tv > c.pos_emb_max_len) return fail(h, -1, "%d encoder frames exceed pos_emb_max_len %d", Tv, c.pos_emb_max_len);
  if (Tv <= 0) return fail(h, -1, "no encoder frames");
  const bool conv2d = c.subsampling == GAM_SUBS_CONV2D;
  // Ta: per-utterance row stride of every token-major buffer (>= T' ; the extra rows are
  // treated as padding).  conv2d: image rows per utterance = 2*Ta >= T1 + 2.
  // conv1d: stage-1 output rows per utterance = 2*Ta >= T1 + 4.
  int Ta = conv2d ? std::max(Tv + 1, (T1 + 3) / 2) : std::max(Tv, (T1 + 5) / 2);
  const int N = B * Ta;
  // Packed rows (gam_pack.h): with the lengths known on the HOST (gam_encode_varlen) a ragged batch runs its layers on the valid frames
  // only -- NR rows instead of N, utterance b at rows cu[b] .. -- when that drops at least 3 % of the rows.  Equal-length batches, single
  // utterances and callers without host lengths keep the padded layout (and its results, bit for bit).
  bool packed = false;
  int NR = N, Tmax = Ta;
  // (below graph_max_rows the step is launch-bound and replayed as a hipGraph keyed on the padded shape: fewer rows buy nothing there and a
  //  row count in the key would defeat the replay -- packed rows start where the graph regime ends; GAM_PACK=2 forces them at any size: tests)
  const bool pack_regime = h->use_pack >= 2 || !h->use_graph || N >= h->graph_max_rows;
  if (feat_len_host != nullptr && h->use_pack && pack_regime && B > 1 && B <= 1024) {
    long long sum = 0;
    int mx = 0;
    for (int b = 0; b < B; ++b) {
      const int64_t l = std::min<int64_t>(std::max<int64_t>(feat_len_host[b], 0), T);
      const int l2 = (int)half_up(half_up(l));      // = len2 of gam_lengths_kernel
      sum += l2;
      mx = std::max(mx, l2);
    }
    if (sum > 0 && sum * 100 <= (long long)N * 97) { packed = true; NR = (int)sum; Tmax = mx; }
  }
  h->last_rows = NR;
  h->last_rows_padded = N;

  if (B > h->lens_cap) {
    if (h->lens) HIPCHK(h, hipFree(h->lens));
    h->lens = nullptr;
    HIPCHK(h, hipMalloc(&h->lens, (size_t)4 * B * sizeof(int)));
    h->lens_cap = B;
    ++h->ws_generation;   // captured graphs hold the old pointer
  }
  int *len0 = h->lens, *len1 = h->lens + h->lens_cap, *len2 = h->lens + 2 * h->lens_cap, *elen = h->lens + 3 * h->lens_cap;
  if (int r = ensure(h, h->x, (size_t)N * D)) return r;
  if (int r = ensure(h, h->y, (size_t)N * D)) return r;
  if (int r = ensure(h, h->yr, (size_t)N * D)) return r;
  if (int r = ensure(h, h->hbuf, (size_t)N * DFF)) return r;
  if (int r = ensure(h, h->qkv, (size_t)N * 3 * D)) return r;
  if (int r = ensure(h, h->ctx, (size_t)N * D)) return r;
  if (int r = ensure(h, h->ubuf, (size_t)N * 2 * D)) return r;
  if (int r = ensure(h, h->zbuf, (size_t)N * D)) return r;
  if (int r = ensure(h, h->rsbuf, (size_t)N + 64)) return r;
  if (packed)
    if (int r = ensure(h, h->pack_idx, (size_t)1032 + 2 * (size_t)N)) return r;
  int* const cu = packed ? reinterpret_cast<int*>(h->pack_idx.p) : nullptr;
  int* const row_t = packed ? cu + 1032 : nullptr;
  int* const row_src = packed ? row_t + N : nullptr;

  {
    ProfScope ps(h, s, GAM_PF_MISC, 0.0);
    hipLaunchKernelGGL(gam_lengths_kernel, dim3(gam_cdiv(B, 64)), dim3(64), 0, s, (const long long*)feat_len, B, (int)T, 2,
                       len0, len1, len2, elen);
    HIPCHK(h, hipGetLastError());
    if (packed) {
      hipLaunchKernelGGL(gam_pack_index_kernel, dim3(1), dim3(256), 0, s, len2, B, Ta, NR, cu, row_t, row_src, h->range_flag);
      HIPCHK(h, hipGetLastError());
    }
  }

  // Large batches: every big GEMM runs on the LDS-DMA kernel of gam_gemm_sp.h, and the kernels that
  // produce its A operands (LayerNorm, SiLU epilogue, attention, conv module, stem) write them in the
  // sp32 split layout instead of fp32 -- same bytes, no conversion pass.
  const bool sp = split_mode(h) && h->use_sp && N >= h->sp_min_m && D % 32 == 0 && DFF % 32 == 0;
  // operand format the producers write (gam_common.h gam_store4).  Format 2 (plain fp16, GAM_GEMM_F16) packs 64 k-values into
  // the 128-byte line of a k-tile, so every reduction length / row pitch must be a multiple of 64; a model whose d_model or
  // FFN width is 32 x an odd number keeps the three-term kernels for its GEMMs in that mode (as gam_op_gemm does)
  // instead of failing half-way through a batch with the stem image already written as fp16.
  const bool f16_fmt_ok = D % 64 == 0 && DFF % 64 == 0;   // (C == D: the stem's channel count)
  const int spf = sp ? (h->gemm_mode == GAM_GEMM_F16 && f16_fmt_ok ? 2 : 1) : 0;
  auto sp_a = [&](GamGemmArgs& g) { if (sp) { g.Asp = reinterpret_cast<const _Float16*>(g.A); g.a_fmt = spf; } };

  // ------------------------------ stem ------------------------------
  if (conv2d) {
    const int FP = h->f1 + 1, F2 = h->f2;
    if (int r = ensure(h, h->img, ((size_t)B * 2 * Ta + 2) * FP * C)) return r;
    if (int r = ensure(h, h->c2, (size_t)N * F2 * C)) return r;
    {
      GamConv1Args a;
      a.feat = feat; a.img = h->img.p; a.w = h->c1_w; a.bias = h->c1_b; a.len0 = len0; a.len1 = len1;
      a.B = B; a.T = (int)T; a.F = F; a.Ta = Ta; a.FP = FP; a.C = C; a.T1 = T1;
      a.img_split = (sp && C % 32 == 0) ? spf : 0;
      a.range_flag = h->use_range ? h->range_flag : nullptr;
      ProfScope ps(h, s, GAM_PF_STEM, (double)B * 2 * Ta * FP * C * 4.0);
      if (a.img_split == 2) hipLaunchKernelGGL(gam_conv2d1_kernel<2>, dim3(2 * Ta, B), dim3(256), 0, s, a);
      else if (a.img_split == 1) hipLaunchKernelGGL(gam_conv2d1_kernel<1>, dim3(2 * Ta, B), dim3(256), 0, s, a);
      else hipLaunchKernelGGL(gam_conv2d1_kernel<0>, dim3(2 * Ta, B), dim3(256), 0, s, a);
      HIPCHK(h, hipGetLastError());
      // two slack rows past the last utterance (read by its padding frame only)
      // two slack rows past the last utterance (read by its padding frame only); the dense fp16 image ends at half the offset
      const size_t img_end = (size_t)B * 2 * Ta * FP * C;
      HIPCHK(h, hipMemsetAsync(a.img_split == 2 ? reinterpret_cast<float*>(reinterpret_cast<_Float16*>(h->img.p) + img_end) : h->img.p + img_end, 0,
                               (size_t)2 * FP * C * sizeof(float), s));
    }
    GamGemmArgs g = gemm_args(h->img.p, 0, h->c2_w, h->c2_b, h->c2.p, C, N * F2, C, 9 * C);
    g.a_mode = 1; g.conv_fp = FP; g.conv_c = C; g.conv_f2 = F2;
    g.lens = len2; g.rpb = Ta * F2; g.fdiv = F2;
    g.skip_pad = packed ? 1 : 0;      // 16-frame row tiles behind an utterance's last frame: 12 rounds of tiles shrink with the batch's padding
    if (sp && C % 32 == 0) { sp_a(g); g.c_split = spf; }
    g.c_guard = 1;
    if (int r = gemm(h, s, g, GAM_ACT_RELU, GAM_PF_CONV2, &h->s_c2)) return r;
    GamGemmArgs l = gemm_args(h->c2.p, (long)F2 * C, h->lin_w, h->lin_b, packed ? h->y.p : h->x.p, D, N, D, F2 * C);
    if (sp && C % 32 == 0) sp_a(l);
    if (int r = gemm(h, s, l, GAM_ACT_NONE, GAM_PF_GEMM, &h->s_lin)) return r;
  } else {
    const int ks = c.subs_kernel_size, pad = (ks - 1) / 2;
    // stage-1 input rows per utterance: 2*T1a >= T + 2*pad ; stage-1 output = stage-2 input rows: 2*Ta
    const int T1a = std::max(T1, (int)((T + 2 * pad + 1) / 2));
    if (int r = ensure(h, h->xin, ((size_t)B * 2 * T1a + 8) * F)) return r;
    if (int r = ensure(h, h->y1, ((size_t)B * 2 * Ta + 8) * C)) return r;
    {
      ProfScope ps(h, s, GAM_PF_STEM, (double)B * T * F * 8.0);
      HIPCHK(h, hipMemsetAsync(h->xin.p, 0, ((size_t)B * 2 * T1a + 8) * F * sizeof(float), s));
      HIPCHK(h, hipMemsetAsync(h->y1.p, 0, ((size_t)B * 2 * Ta + 8) * C * sizeof(float), s));
      dim3 grid(gam_cdiv(T, 32), gam_cdiv(F, 32), B);
      hipLaunchKernelGGL(gam_feat_to_rows_kernel, grid, dim3(256), 0, s, feat, h->xin.p, len0, B, F, (int)T, 2 * T1a, pad);
      HIPCHK(h, hipGetLastError());
    }
    GamGemmArgs g1 = gemm_args(h->xin.p, 2L * F, h->c1_w, h->c1_b, h->y1.p, C, B * T1a, C, ks * F);
    g1.lens = len1; g1.rpb = T1a; g1.fdiv = 1;
    g1.remap = 1; g1.out_rpb = 2 * Ta; g1.out_shift = pad; g1.rows_valid = std::min(T1a, 2 * Ta - pad);
    if (int r = gemm(h, s, g1, GAM_ACT_RELU, GAM_PF_CONV2, &h->s_c1)) return r;
    GamGemmArgs g2 = gemm_args(h->y1.p, 2L * C, h->c2_w, h->c2_b, packed ? h->y.p : h->x.p, D, N, D, ks * C);
    g2.lens = len2; g2.rpb = Ta; g2.fdiv = 1;
    if (int r = gemm(h, s, g2, GAM_ACT_RELU, GAM_PF_CONV2, &h->s_c2)) return r;
  }
  if (packed) {   // the stem's padded rows -> packed rows (the only copy the layout costs: 2 x NR x D x 4 bytes)
    ProfScope ps(h, s, GAM_PF_STEM, (double)NR * D * 8.0);
    hipLaunchKernelGGL(gam_gather_rows_kernel, dim3(std::min(4096, gam_cdiv(NR * (D / 4), 256))), dim3(256), 0, s, h->y.p, row_src, h->x.p, NR, D);
    HIPCHK(h, hipGetLastError());
  }

  // ------------------------------ Conformer layers ------------------------------
  const int nl = n_layers_run < 0 ? c.n_layers : std::min(n_layers_run, c.n_layers);
  const bool rel = c.self_attention_model == GAM_ATT_REL_POS;
  if (rel) {
    if (int r = ensure(h, h->pbuf, (size_t)(2 * Tv - 1) * D)) return r;
  }
  auto run_layers = [&](hipStream_t ls) -> int {
  hipStream_t s = ls;   // (shadows the caller's stream: the capture runs on a private one)
  GamLnArgs ln;
  memset(&ln, 0, sizeof ln);
  ln.rows = NR; ln.d = D; ln.ta = Ta; ln.row_t = row_t; ln.dk = dk; ln.eps = 1e-5f; ln.rcos = h->rot_cos; ln.rsin = h->rot_sin;
  ln.rope_rows = c.pos_emb_max_len;
  // split-fp16 modes: every LayerNorm hands its GEMMs a per-row power-of-two scale (gam_row_scale)
  float* const rs = split_mode(h) && h->use_rowscale ? h->rsbuf.p : nullptr;
  ln.rs = rs;
  if (nl > 0) {
    GamLnArgs a = ln;
    a.x = h->x.p; a.out1 = h->y.p; a.w1 = h->layers[0].ln_ff1_w; a.b1 = h->layers[0].ln_ff1_b;
    a.split1 = spf;
    if (int r = layernorm(h, s, a, 0)) return r;
  }
  PendingReduce pend;    // the split-K slices of the residual GEMM just launched, summed by the LayerNorm that follows it
  for (int li = 0; li < nl; ++li) {
    const LayerW& L = h->layers[li];
    // --- FFN 1 (macaron half step) ---
    {
      GamGemmArgs g = gemm_args(h->y.p, D, L.ff1_w1, L.ff1_b1, h->hbuf.p, DFF, NR, DFF, D);
      sp_a(g); g.c_split = spf; g.a_rs = rs; g.c_guard = 1;
      if (int r = gemm(h, s, g, GAM_ACT_SILU, GAM_PF_GEMM, &L.s_ff1_w1)) return r;
      GamGemmArgs g2 = gemm_args(h->hbuf.p, DFF, L.ff1_w2, L.ff1_b2, h->x.p, D, NR, D, DFF);
      g2.R = h->x.p; g2.ldr = D; g2.alpha = 0.5f;
      sp_a(g2);
      if (int r = gemm(h, s, g2, GAM_ACT_NONE, GAM_PF_GEMM, &L.s_ff1_w2, &pend)) return r;
    }
    // --- self attention ---
    {
      GamLnArgs a = ln;
      a.x = h->x.p; a.out1 = h->y.p; a.out2 = h->yr.p; a.w1 = L.ln_att_w; a.b1 = L.ln_att_b;
      a.split1 = spf; a.split2 = spf;
      if (int r = layernorm(h, s, a, rel ? 0 : 1, &pend)) return r;
      // rotary: q,k project the rotated copy, v the plain one; rel_pos: all three project y.  One [N, 3D] result q | k | v.
      // q, k and v are split to fp16 unscaled by the attention kernel: range guard on all of them.
      const bool one_launch = rel || (sp && (2 * D) % 256 == 0);   // (the operand switch sits on a tile boundary of either tile width)
      if (one_launch) {
        GamGemmArgs gq = gemm_args(rel ? h->y.p : h->yr.p, D, L.wqkv, L.bqkv, h->qkv.p, 3 * D, NR, 3 * D, D);
        sp_a(gq); gq.a_rs = rs; gq.c_guard = 1;
        if (!rel) { gq.Asp2 = reinterpret_cast<const _Float16*>(h->y.p); gq.n_switch = 2 * D; }
        if (int r = gemm(h, s, gq, GAM_ACT_NONE, GAM_PF_GEMM, &L.s_wqkv)) return r;
      } else {
        GamGemmArgs gq = gemm_args(h->yr.p, D, L.wqkv, L.bqkv, h->qkv.p, 3 * D, NR, 2 * D, D);
        sp_a(gq); gq.a_rs = rs; gq.c_guard = 1;
        if (int r = gemm(h, s, gq, GAM_ACT_NONE, GAM_PF_GEMM, &L.s_wqkv)) return r;
        W16 wv16 = L.s_wqkv;   // rows 2D .. 3D of the same planes
        const size_t off = (size_t)2 * D * D;
        if (wv16.hi) { wv16.hi += off; wv16.lo += off; }
        if (wv16.sp) wv16.sp += 2 * off;
        GamGemmArgs gv = gemm_args(h->y.p, D, L.wqkv + off, L.bqkv + 2 * D, h->qkv.p + 2 * D, 3 * D, NR, D, D);
        sp_a(gv); gv.a_rs = rs; gv.c_guard = 1;
        if (int r = gemm(h, s, gv, GAM_ACT_NONE, GAM_PF_GEMM, &wv16)) return r;
      }
      if (rel) {  // P = linear_pos(pos_emb) for relative positions -(T'-1) .. T'-1 (no bias)
        const float* pe0 = h->rel_pe + (size_t)(c.pos_emb_max_len - 1 - (Tv - 1)) * D;
        GamGemmArgs gp = gemm_args(pe0, D, L.wpos, nullptr, h->pbuf.p, D, 2 * Tv - 1, D, D);
        if (int r = gemm(h, s, gp, GAM_ACT_NONE, GAM_PF_GEMM, &L.s_wpos)) return r;
      }
      GamAttnArgs at;
      memset(&at, 0, sizeof at);
      at.q = h->qkv.p; at.k = h->qkv.p + D; at.v = h->qkv.p + 2 * D; at.ctx = h->ctx.p; at.ctx_split = spf;
      at.lens = B > 1 ? len2 : nullptr;  // encoder.py:620-624: no mask at batch 1
      at.cu = cu; at.B = B; at.Ta = packed ? Tmax : Ta; at.Tv = Tv; at.H = H; at.ldq = 3 * D; at.ldv = 3 * D; at.ldo = D;
      at.scale = 1.0f / sqrtf((float)dk);
      at.pbuf = rel ? h->pbuf.p : nullptr; at.pos_u = L.pos_u; at.pos_v = L.pos_v; at.ldp = D;
      {
        ProfScope ps(h, s, GAM_PF_ATTN, 4.0 * (double)B * H * (double)Tv * Tv * dk);
        hipError_t e = gam_launch_attn_mode(at, dk, split_mode(h), s, h->gemm_mode == GAM_GEMM_F16 ? 1 : 3, h->ncu);
        if (e != hipSuccess) return fail(h, -2, "attention launch: %s", hipGetErrorString(e));
      }
      GamGemmArgs go = gemm_args(h->ctx.p, D, L.wo, L.bo, h->x.p, D, NR, D, D);
      go.R = h->x.p; go.ldr = D;
      sp_a(go);
      if (int r = gemm(h, s, go, GAM_ACT_NONE, GAM_PF_GEMM, &L.s_wo, &pend)) return r;
    }
    // --- convolution module ---
    {
      GamLnArgs a = ln;
      a.x = h->x.p; a.out1 = h->y.p; a.w1 = L.ln_conv_w; a.b1 = L.ln_conv_b;
      a.split1 = spf;
      if (int r = layernorm(h, s, a, 0, &pend)) return r;
      GamGemmArgs g1 = gemm_args(h->y.p, D, L.pw1_w, L.pw1_b, h->ubuf.p, 2 * D, NR, 2 * D, D);
      sp_a(g1); g1.a_rs = rs;
      if (int r = gemm(h, s, g1, GAM_ACT_NONE, GAM_PF_GEMM, &L.s_pw1)) return r;
      GamConvModArgs cm;
      cm.u = h->ubuf.p; cm.z = h->zbuf.p; cm.dw_w = L.dw_w; cm.dw_b = L.dw_b; cm.n_scale = L.cn_scale; cm.n_shift = L.cn_shift;
      cm.lens = len2; cm.cu = cu; cm.B = B; cm.Ta = packed ? Tmax : Ta; cm.Tv = Tv; cm.d = D; cm.ks = c.conv_kernel_size; cm.eps = 1e-5f;
      cm.z_split = spf; cm.range_flag = h->use_range ? h->range_flag : nullptr;
      {
        ProfScope ps(h, s, GAM_PF_CONVMOD, (double)NR * D * 3 * 4.0);
        hipError_t e = gam_launch_convmod(cm, c.conv_norm_type == GAM_NORM_LAYER, s);
        if (e != hipSuccess) return fail(h, -2, "conv-module launch (k=%d): %s", cm.ks, hipGetErrorString(e));
      }
      GamGemmArgs g2 = gemm_args(h->zbuf.p, D, L.pw2_w, L.pw2_b, h->x.p, D, NR, D, D);
      g2.R = h->x.p; g2.ldr = D;
      sp_a(g2);
      if (int r = gemm(h, s, g2, GAM_ACT_NONE, GAM_PF_GEMM, &L.s_pw2, &pend)) return r;
    }
    // --- FFN 2 ---
    {
      GamLnArgs a = ln;
      a.x = h->x.p; a.out1 = h->y.p; a.w1 = L.ln_ff2_w; a.b1 = L.ln_ff2_b;
      a.split1 = spf;
      if (int r = layernorm(h, s, a, 0, &pend)) return r;
      GamGemmArgs g = gemm_args(h->y.p, D, L.ff2_w1, L.ff2_b1, h->hbuf.p, DFF, NR, DFF, D);
      sp_a(g); g.c_split = spf; g.a_rs = rs; g.c_guard = 1;
      if (int r = gemm(h, s, g, GAM_ACT_SILU, GAM_PF_GEMM, &L.s_ff2_w1)) return r;
      GamGemmArgs g2 = gemm_args(h->hbuf.p, DFF, L.ff2_w2, L.ff2_b2, h->x.p, D, NR, D, DFF);
      g2.R = h->x.p; g2.ldr = D; g2.alpha = 0.5f;
      sp_a(g2);
      if (int r = gemm(h, s, g2, GAM_ACT_NONE, GAM_PF_GEMM, &L.s_ff2_w2, &pend)) return r;
    }
    // --- norm_out (+ next layer's norm_feed_forward1) ---
    {
      GamLnArgs a = ln;
      a.x = h->x.p; a.out1 = h->x.p; a.w1 = L.ln_out_w; a.b1 = L.ln_out_b;
      if (li + 1 < nl) {
        a.out2 = h->y.p; a.w2 = h->layers[li + 1].ln_ff1_w; a.b2 = h->layers[li + 1].ln_ff1_b;
        a.split2 = spf;
        if (int r = layernorm(h, s, a, 2, &pend)) return r;
      } else {
        a.rs = nullptr;   // the last norm_out feeds no GEMM
        if (int r = layernorm(h, s, a, 0, &pend)) return r;
      }
    }
  }
  return flush_pending(h, s, &pend);   // (nothing is pending after a norm_out; kept so that a future reordering cannot lose a reduce)
  };

  // Small batches: replay the layer sequence as one hipGraph (see gam_handle::use_graph).
  bool replayed = false;
  if (h->use_graph && !h->prof_on && nl > 0 && N < h->graph_max_rows && !packed) {   // (packed rows: the row count is part of every launch)
    // every workspace the captured launches touch must exist before the capture (no allocation inside)
    if (int r = ensure(h, h->splitk_ws, (size_t)16 * N * std::max(DFF, 2 * D) + 64)) return r;
    const GamSpForce& frc = gam_sp_force();   // (a plan forced through gam_tune_sp after a capture must not replay the old tiling)
    const std::vector<int> key = {B, Ta, Tv, nl, h->gemm_mode, (int)sp, h->use_splitk, h->fuse_reduce, frc.mt.load(), frc.nw.load(), frc.s.load(), frc.ns.load()};
    gam_handle::GraphEntry& ge = h->graphs[key];
    if (ge.gen != h->ws_generation) {   // a buffer moved since this entry was made
      if (ge.exec) { hipGraphExecDestroy(ge.exec); --h->graph_count; }
      ge = gam_handle::GraphEntry();
      ge.gen = h->ws_generation;
    }
    if (ge.exec == nullptr && ge.seen == 1 && h->graph_count >= 32) ge.seen = 2;   // cache full: plain launches
    if (ge.exec == nullptr && ge.seen == 1) {
      hipGraph_t graph = nullptr;
      if (h->cap_stream == nullptr && hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking) != hipSuccess)
        h->cap_stream = nullptr;
      if (h->cap_stream != nullptr && hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeRelaxed) == hipSuccess) {
        const int r = run_layers(h->cap_stream);     // recorded, not executed
        const hipError_t e = hipStreamEndCapture(h->cap_stream, &graph);
        if (r == 0 && e == hipSuccess && graph != nullptr && h->ws_generation == ge.gen &&
            hipGraphInstantiate(&ge.exec, graph, nullptr, nullptr, 0) != hipSuccess)
          ge.exec = nullptr;
        if (graph) hipGraphDestroy(graph);
      }
      (void)hipGetLastError();               // a refused capture must not poison the launches below
      if (ge.exec == nullptr) { ge.seen = 2; ++h->graph_captures_failed; }
      else ++h->graph_count;   // capture failed: never try this shape again
    }
    if (ge.exec != nullptr) {
      HIPCHK(h, hipGraphLaunch(ge.exec, s));
      replayed = true;
      ++h->graph_replays;
    } else if (ge.seen == 0) {
      ge.seen = 1;
    }
  }
  if (!replayed)
    if (int r = run_layers(s)) return r;

  // ------------------------------ outputs ------------------------------
  {
    ProfScope ps(h, s, GAM_PF_MISC, (double)B * Tv * D * 8.0);
    if (encoded && packed) {
      hipLaunchKernelGGL(gam_unpack_transpose_kernel, dim3(gam_cdiv(D, 32), gam_cdiv(Tv, 32), B), dim3(256), 0, s, h->x.p, cu, len2, encoded, Tv, D);
      HIPCHK(h, hipGetLastError());
    } else if (encoded) {
      hipError_t e = gam_launch_transpose(h->x.p, encoded, B, Tv, D, (size_t)Ta * D, D, (size_t)D * Tv, Tv, s);
      if (e != hipSuccess) return fail(h, -2, "transpose launch: %s", hipGetErrorString(e));
    }
    if (tokens_out && packed) {
      hipLaunchKernelGGL(gam_unpack_rows_kernel, dim3(std::min(4096, gam_cdiv(B * Tv * (D / 4), 256))), dim3(256), 0, s, h->x.p, cu, len2, tokens_out, B, Tv, D);
      HIPCHK(h, hipGetLastError());
    } else if (tokens_out)
      HIPCHK(h, hipMemcpy2DAsync(tokens_out, (size_t)Tv * D * 4, h->x.p, (size_t)Ta * D * 4, (size_t)Tv * D * 4, B, hipMemcpyDeviceToDevice, s));
    if (enc_len) HIPCHK(h, hipMemcpyAsync(enc_len, elen, (size_t)B * sizeof(int), hipMemcpyDeviceToDevice, s));
  }
  return 0;
}

int gam_encode(gam_handle* h, const float* feat, const int64_t* feat_len, int B, int64_t T, float* encoded, int32_t* enc_len,
               void* stream) {
  return encode_impl(h, feat, feat_len, B, T, encoded, enc_len, -1, nullptr, (hipStream_t)stream);
}

int gam_encode_ex(gam_handle* h, const float* feat, const int64_t* feat_len, int B, int64_t T, float* encoded, int32_t* enc_len,
                  int n_layers_run, float* tokens_out, void* stream) {
  return encode_impl(h, feat, feat_len, B, T, encoded, enc_len, n_layers_run, tokens_out, (hipStream_t)stream);
}

int gam_encode_varlen(gam_handle* h, const float* feat, const int64_t* feat_len, const int64_t* feat_len_host, int B, int64_t T,
                      float* encoded, int32_t* enc_len, int n_layers_run, float* tokens_out, void* stream) {
  return encode_impl(h, feat, feat_len, B, T, encoded, enc_len, n_layers_run, tokens_out, (hipStream_t)stream, feat_len_host);
}

// -----------------------------------------------------------------------------------
// heads: encoded [B,D,Tp] -> token-major [B*Tp, D] once, then GEMMs
static int to_tokens(gam_handle* h, const float* encoded, int B, int64_t Tp, hipStream_t s) {
  const int D = h->cfg.d_model;
  if (int r = ensure(h, h->tok, (size_t)B * Tp * D)) return r;
  ProfScope ps(h, s, GAM_PF_DECODE, (double)B * Tp * D * 8.0);
  hipError_t e = gam_launch_transpose(encoded, h->tok.p, B, D, (int)Tp, (size_t)D * Tp, Tp, (size_t)Tp * D, D, s);
  if (e != hipSuccess) return fail(h, -2, "transpose launch: %s", hipGetErrorString(e));
  return 0;
}

static int ctc_logits(gam_handle* h, const float* encoded, int B, int64_t Tp, hipStream_t s) {
  if (!h || !h->finalized) return fail(h, -1, "CTC head before gam_finalize");
  if (h->cfg.head_type != GAM_HEAD_CTC || !h->has_head) return fail(h, -1, "model has no CTC head");
  if (B <= 0 || Tp <= 0) return fail(h, -1, "bad shape B=%d T'=%lld", B, (long long)Tp);
  HIPCHK(h, hipSetDevice(h->device));
  const int D = h->cfg.d_model, V = h->cfg.num_classes;
  if (int r = to_tokens(h, encoded, B, Tp, s)) return r;
  if (int r = ensure(h, h->logits, (size_t)B * Tp * V)) return r;
  GamGemmArgs g = gemm_args(h->tok.p, D, h->ctc_w, h->ctc_b, h->logits.p, V, (int)(B * Tp), V, D);
  return gemm(h, s, g, GAM_ACT_NONE, GAM_PF_DECODE);
}

int gam_ctc_head(gam_handle* h, const float* encoded, int B, int64_t Tp, float* log_probs, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!h) return -1;
  HIPCHK(h, hipSetDevice(h->device));     // (before the scope: its event belongs to the handle's device)
  DecodeScope ds(h, s);
  if (int r = ctc_logits(h, encoded, B, Tp, s)) return r;
  const int V = h->cfg.num_classes, rows = (int)(B * Tp);
  ProfScope ps(h, s, GAM_PF_DECODE, (double)rows * V * 8.0);
  hipLaunchKernelGGL(gam_log_softmax_kernel, dim3(gam_cdiv(rows, 4)), dim3(256), 0, s, h->logits.p, log_probs, rows, V);
  HIPCHK(h, hipGetLastError());
  return 0;
}

int gam_ctc_greedy(gam_handle* h, const float* encoded, const int32_t* enc_len, int B, int64_t Tp, int32_t* ids,
                   int32_t* frames, int32_t* counts, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!h) return -1;
  HIPCHK(h, hipSetDevice(h->device));     // (before the scope: its event belongs to the handle's device)
  DecodeScope ds(h, s);
  if (int r = ctc_logits(h, encoded, B, Tp, s)) return r;
  const int V = h->cfg.num_classes;
  const size_t sm = ((size_t)Tp + GAM_CTC_NT / 64 + 8) * sizeof(int);
  if (sm > 60 * 1024) return fail(h, -1, "T'=%lld too long for the CTC greedy kernel", (long long)Tp);
  ProfScope ps(h, s, GAM_PF_DECODE, (double)B * Tp * V * 4.0);
  hipLaunchKernelGGL(gam_ctc_greedy_kernel, dim3(B), dim3(GAM_CTC_NT), sm, s, h->logits.p, enc_len, (int)Tp, V, ids, frames, counts);
  HIPCHK(h, hipGetLastError());
  return 0;
}

#if GAM_RC_AUDIT
// diagnosis build: print what the PREVIOUS cluster decode's audit logged (called before the next decode reuses the log, and at destroy --
// never right behind the launch: a host sync there would remove the very overlap under test)
static int audit_dump(gam_handle* h) {
  if (h->rnnt_audit.p == nullptr || !h->audit_pending) return 0;
  h->audit_pending = false;
  static std::vector<int> au(16 + 12 * GAM_RC_AUDIT_CAP);
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemcpy(au.data(), h->rnnt_audit.p, 64, hipMemcpyDeviceToHost));
  const int n = au[0];
  h->audit_decodes++;
  if (n > 0) {
    h->audit_hit_decodes++;
    const int m = n < GAM_RC_AUDIT_CAP ? n : GAM_RC_AUDIT_CAP;
    HIPCHK(h, hipMemcpy(au.data(), h->rnnt_audit.p, (16 + 12 * (size_t)m) * sizeof(int), hipMemcpyDeviceToHost));
    int by_site[16] = {0};
    for (int k = 0; k < m; ++k) by_site[au[16 + 12 * k] & 15]++;
    fprintf(stderr, "[gam-audit] decode %ld C=%d entries %d; by site:", h->audit_decodes, h->audit_last_c, n);
    for (int k = 1; k < 16; ++k) if (by_site[k]) fprintf(stderr, " s%d=%d", k, by_site[k]);
    fprintf(stderr, "\n");
    for (int k = 0; k < m && k < 24; ++k) {
      const int* e = au.data() + 16 + 12 * k;
      float x0, x1; memcpy(&x0, e + 6, 4); memcpy(&x1, e + 7, 4);
      fprintf(stderr, "[gam-audit]   site %d utt %d member %d xc %d tid %d idx %d: %.9g (%08x) vs %.9g (%08x)  aux %08x %08x\n", e[0], e[1], e[2], e[3], e[4], e[5],
              x0, (unsigned)e[6], x1, (unsigned)e[7], (unsigned)e[8], (unsigned)e[9]);
    }
  }
  return 0;
}
#endif

int gam_rnnt_greedy(gam_handle* h, const float* encoded, const int32_t* enc_len, int B, int64_t Tp, int max_symbols,
                    int32_t* ids, int32_t* frames, int32_t* counts, float* logits_dump, int32_t* dump_count, int dump_cap,
                    void* stream) {
  if (!h || !h->finalized) return fail(h, -1, "RNN-T head before gam_finalize");
  if (h->cfg.head_type != GAM_HEAD_RNNT || !h->has_head) return fail(h, -1, "model has no RNN-T head");
  if (B <= 0 || Tp <= 0 || max_symbols <= 0) return fail(h, -1, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(h, hipSetDevice(h->device));
  DecodeScope ds(h, s);
  const gam_config& c = h->cfg;
  const int D = c.d_model, JH = c.joint_hidden;
  if (int r = to_tokens(h, encoded, B, Tp, s)) return r;
  if (int r = ensure(h, h->encp, (size_t)B * Tp * JH)) return r;
  GamGemmArgs g = gemm_args(h->tok.p, D, h->jn_enc_w, h->jn_enc_b, h->encp.p, JH, (int)(B * Tp), JH, D);
  if (int r = gemm(h, s, g, GAM_ACT_NONE, GAM_PF_DECODE)) return r;
  GamRnntArgs a;
  memset(&a, 0, sizeof a);
  a.encp = h->encp.p; a.enc_len = enc_len; a.gate_tab = h->lstm_tab; a.whh_t = h->lstm_whh_t; a.wpred_t = h->jn_pred_t;
  a.bpred = h->jn_pred_b; a.wout = h->jn_out_w; a.bout = h->jn_out_b; a.ids = ids; a.frames = frames; a.counts = counts;
  a.dump = logits_dump; a.dump_count = dump_count; a.B = B; a.Tp = (int)Tp; a.V = c.num_classes; a.H = c.pred_hidden; a.JH = JH;
  a.max_symbols = max_symbols; a.cap = (int)Tp * max_symbols; a.dump_cap = logits_dump ? dump_cap : 0;
  a.L = c.pred_rnn_layers; a.wih_x = h->lstm_wih_x; a.whh_x = h->lstm_whh_x; a.bias_x = h->lstm_bias_x;
  ProfScope ps(h, s, GAM_PF_DECODE, 0.0);
  // One workgroup per utterance (gam_decode.h): the whole decode when the cluster kernel does not take the shape
  // (GAM_RNNT_CLUSTER=0, a refused cooperative launch), and -- with only_failed -- the REPAIR pass behind every cluster
  // launch: a workgroup of it returns at once unless the cluster kernel left its utterance at counts[b] = -1 (a hand-off
  // timed out because a member was not resident: GPU shared with another process / stream, CU-masked partition).  The
  // caller therefore always gets the reference's ids back, without a host round trip and without an exception.
  auto launch_single = [&](int only_failed) -> int {
    GamRnntArgs f = a;
    f.only_failed = only_failed;
    f.wout_in_lds = gam_rnnt_smem(f.H, f.JH, f.V, 1, f.L) <= 96 * 1024 ? 1 : 0;
    size_t sm1 = gam_rnnt_smem(f.H, f.JH, f.V, f.wout_in_lds, f.L);
    if (h->rnnt_exclusive && B <= h->ncu && sm1 <= 160 * 1024) sm1 = 160 * 1024;   // (owns its CU like a cluster member: see below)
    static std::atomic<unsigned long long> attr5{0}, attr8{0};
    HIPCHK(h, gam_set_max_lds(reinterpret_cast<const void*>(gam_rnnt_greedy_kernel<5>), 160 * 1024, attr5));
    HIPCHK(h, gam_set_max_lds(reinterpret_cast<const void*>(gam_rnnt_greedy_kernel<8>), 160 * 1024, attr8));
    if (sm1 > 160 * 1024) return fail(h, -1, "RNN-T head too large for the greedy kernel's LDS window");
    if (4 * f.H <= 256 * 5) hipLaunchKernelGGL(gam_rnnt_greedy_kernel<5>, dim3(B), dim3(256), sm1, s, f);
    else hipLaunchKernelGGL(gam_rnnt_greedy_kernel<8>, dim3(B), dim3(256), sm1, s, f);
    HIPCHK(h, hipGetLastError());
    return 0;
  };
  // Cluster decode (gam_decode_cluster.h): C workgroups per utterance, grid <= one workgroup per CU.
  // (8 utterance columns, one per XCD; C members of a cluster share an XCD.)
  {
    const int nu8 = gam_cdiv(B, 8);
    // every member of every cluster must be resident at once (they spin on each other): at most one workgroup per CU
    // of THIS device, with a few CUs to spare; a partition with fewer CUs gets smaller clusters or the one-workgroup kernel
    int C = (h->ncu - 16) / (8 * nu8);
    C = C > 8 ? 8 : C;
    if (h->rnnt_cluster >= 0) C = h->rnnt_cluster < C ? h->rnnt_cluster : C;
    if (C >= 1 && JH % 16 == 0 && a.H % 4 == 0 && a.L == 1) {   // (JH % 16: MFMA k-steps and the LDS-DMA window; L > 1: one-workgroup kernel)
      GamRnntClusterArgs ca;
      memset(&ca, 0, sizeof ca);
      ca.a = a; ca.whh_q = h->lstm_whh_q; ca.wpred_q = h->jn_pred_q; ca.C = C;
      ca.force_dead = h->rnnt_force_timeout;
      const int nI = gam_cdiv(a.H, C), need = gam_cdiv(4 * nI, 256);
      int nr = need <= 1 ? 1 : (need <= 2 ? 2 : (need <= 3 ? 3 : (need <= 5 ? 5 : 8)));
      const int nV = gam_cdiv(gam_cdiv(a.V, C), 16) * 16;
      ca.wout_slice_in_lds = (size_t)nV * (JH + 4) * 4 + gam_rnnt_cluster_smem(a.H, JH, a.V, C, nr, 0) <= 96 * 1024 ? 1 : 0;
      ca.wpred_slice_in_lds = gam_rnnt_cluster_smem(a.H, JH, a.V, C, nr, ca.wout_slice_in_lds, 1) <= 150 * 1024 ? 1 : 0;
      size_t sm = gam_rnnt_cluster_smem(a.H, JH, a.V, C, nr, ca.wout_slice_in_lds, ca.wpred_slice_in_lds);
      // GAM_RNNT_EXCLUSIVE=1: a decode workgroup asks for the CU's whole LDS (160 KB), so no other workgroup that uses LDS -- of this launch or
      // of ANOTHER stream's kernel -- is placed beside it.  r05 shipped this as the protection against a perturbation of co-resident decode
      // workgroups whose mechanism was unknown; r06 found it (hipcc had packed the gate rows' FMA chains into v_pk_fma_f32 op_sel:[0,1,0],
      // whose low result is wrong in lanes 48..63 beside another wave's MFMAs: gigaam_amd/build.py) and removed the instruction, so the
      // default is the kernel's own LDS size again; the switch stays as an A/B (profiles/r06_exclusive_ab.txt: no time difference).
      if (h->rnnt_exclusive && sm <= 160 * 1024) sm = 160 * 1024;
      const size_t xg = gam_rnnt_cluster_xgranules(a.H, JH, C) * (size_t)B;
      if (sm <= 160 * 1024 && need <= 8) {
        if (int r = ensure(h, h->rnnt_x, xg * 2 + 64)) return r;   // (floats: 2 per granule) + status word
        HIPCHK(h, hipMemsetAsync(h->rnnt_x.p, 0, (xg * 2 + 64) * sizeof(float), s));
        ca.xbuf = reinterpret_cast<unsigned long long*>(h->rnnt_x.p);
        ca.status = reinterpret_cast<int*>(h->rnnt_x.p + xg * 2);
        const dim3 grid(8 * nu8 * C);
#if GAM_RC_AUDIT
        if (int r = audit_dump(h)) return r;
        h->audit_pending = true; h->audit_last_c = C;
        if (int r = ensure(h, h->rnnt_audit, 16 + 12 * GAM_RC_AUDIT_CAP)) return r;
        HIPCHK(h, hipMemsetAsync(h->rnnt_audit.p, 0, 64, s));
        ca.audit = reinterpret_cast<int*>(h->rnnt_audit.p);
#endif
        // Cooperative launch: the runtime REFUSES a grid that cannot be co-resident on this device (occupancy x CUs)
        // instead of letting resident members spin on absent ones; what it cannot see (CUs held by another queue) is
        // what the bounded spins + the repair pass below are for.
        const bool coop = h->rnnt_coop != 0;
        const void* kern = nullptr;
        static std::atomic<unsigned long long> at1{0}, at2{0}, at3{0}, at5{0}, at8{0}, at1r{0}, at2c{0}, at3c{0}, at5c{0};
        std::atomic<unsigned long long>* at = nullptr;
        const bool resident = nr == 1 && a.H == 320 && JH == 320;   // W_hh rows register-resident (gam_decode_cluster.h RESQ)
        const bool h320 = a.H == 320 && JH == 320;      // compile-time sizes (the published heads)
        if (resident) { kern = reinterpret_cast<const void*>(gam_rnnt_cluster_kernel<1, 80>); at = &at1r; }
        else if (h320 && nr == 2) { kern = reinterpret_cast<const void*>(gam_rnnt_cluster_kernel<2, 0, 320>); at = &at2c; }
        else if (h320 && nr == 3) { kern = reinterpret_cast<const void*>(gam_rnnt_cluster_kernel<3, 0, 320>); at = &at3c; }
        else if (h320 && nr == 5) { kern = reinterpret_cast<const void*>(gam_rnnt_cluster_kernel<5, 0, 320>); at = &at5c; }
        else switch (nr) {
          case 1: kern = reinterpret_cast<const void*>(gam_rnnt_cluster_kernel<1>); at = &at1; break;
          case 2: kern = reinterpret_cast<const void*>(gam_rnnt_cluster_kernel<2>); at = &at2; break;
          case 3: kern = reinterpret_cast<const void*>(gam_rnnt_cluster_kernel<3>); at = &at3; break;
          case 5: kern = reinterpret_cast<const void*>(gam_rnnt_cluster_kernel<5>); at = &at5; break;
          default: kern = reinterpret_cast<const void*>(gam_rnnt_cluster_kernel<8>); at = &at8; break;
        }
        HIPCHK(h, gam_set_max_lds(kern, 160 * 1024, *at));
        void* kargs[] = {&ca};
        hipError_t le = coop ? hipLaunchCooperativeKernel(kern, grid, dim3(256), kargs, (unsigned)sm, s)
                             : hipLaunchKernel(kern, grid, dim3(256), kargs, sm, s);
        if (le != hipSuccess) {
          (void)hipGetLastError();
          static std::atomic<int> warned{0};
          if (!warned.exchange(1))
            fprintf(stderr, "[gam] RNN-T cluster decode: %s launch of %u workgroups refused (%s); decoding with one workgroup per utterance\n",
                    coop ? "cooperative" : "plain", grid.x, hipGetErrorString(le));
          return launch_single(0);
        }
        if (GAM_RC_TIMING && getenv("GAM_RNNT_TIMING")) {
          int st[16];
          HIPCHK(h, hipStreamSynchronize(s));
          HIPCHK(h, hipMemcpy(st, ca.status, sizeof st, hipMemcpyDeviceToHost));
          fprintf(stderr, "[gam] rnnt cluster C=%d%s utt0: rounds %d; us: gates %.0f xH %.0f pred %.0f xP %.0f z %.0f joint %.0f xA %.0f comb %.0f ctrl %.0f\n", C,
                  resident ? " (resident)" : "", st[T_ROUNDS], st[T_GATES] / 100.0, st[T_XH] / 100.0, st[T_PRED] / 100.0, st[T_XP] / 100.0, st[T_Z] / 100.0,
                  st[T_JOINT] / 100.0, st[T_XA] / 100.0, st[T_COMB] / 100.0, st[T_CTRL] / 100.0);
        }
        if (h->rnnt_no_repair) return 0;   // GAM_RNNT_NO_REPAIR=1 (test hook, read at gam_create): leave a failed cluster's counts at -1 (HipEngine.collect then raises)
        return launch_single(1);   // repair pass: no-op workgroups unless a cluster gave up
      }
    }
  }
  return launch_single(0);
}

int gam_rnnt_predict(gam_handle* h, const int32_t* labels, const float* h_in, const float* c_in, int B, float* g_out,
                     float* h_out, float* c_out, void* stream) {
  if (!h || !h->finalized) return fail(h, -1, "gam_rnnt_predict before gam_finalize");
  if (h->cfg.head_type != GAM_HEAD_RNNT || !h->has_head) return fail(h, -1, "model has no RNN-T head");
  if (B <= 0 || !g_out || !h_out || !c_out || (h_in == nullptr) != (c_in == nullptr)) return fail(h, -1, "bad gam_rnnt_predict arguments");
  HIPCHK(h, hipSetDevice(h->device));
  const gam_config& c = h->cfg;
  GamPredictArgs a;
  a.label = labels; a.h_in = h_in; a.c_in = c_in; a.g_out = g_out; a.h_out = h_out; a.c_out = c_out;
  a.gate_tab = h->lstm_tab; a.whh_t = h->lstm_whh_t; a.wih_x = h->lstm_wih_x; a.whh_x = h->lstm_whh_x; a.bias_x = h->lstm_bias_x;
  a.B = B; a.PH = c.pred_hidden; a.V = c.num_classes; a.L = c.pred_rnn_layers;
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(h, s, GAM_PF_DECODE, 0.0);
  hipLaunchKernelGGL(gam_rnnt_predict_kernel, dim3(B), dim3(256), (size_t)6 * c.pred_hidden * sizeof(float), s, a);
  HIPCHK(h, hipGetLastError());
  return 0;
}

int gam_rnnt_joint(gam_handle* h, const float* enc, const float* dec, int B, int T, int U, float* log_probs, void* stream) {
  if (!h || !h->finalized) return fail(h, -1, "gam_rnnt_joint before gam_finalize");
  if (h->cfg.head_type != GAM_HEAD_RNNT || !h->has_head) return fail(h, -1, "model has no RNN-T head");
  if (B <= 0 || T <= 0 || U <= 0 || !enc || !dec || !log_probs) return fail(h, -1, "bad gam_rnnt_joint arguments");
  HIPCHK(h, hipSetDevice(h->device));
  const gam_config& c = h->cfg;
  const int D = c.d_model, PH = c.pred_hidden, JH = c.joint_hidden, V = c.num_classes;
  const size_t rows = (size_t)B * T * U;
  if (rows * std::max(JH, V) > ((size_t)1 << 31)) return fail(h, -1, "gam_rnnt_joint: B x T x U = %zu rows is too large for one call", rows);
  hipStream_t s = (hipStream_t)stream;
  DecodeScope ds(h, s);
  if (int r = ensure(h, h->encp, (size_t)B * T * JH)) return r;
  if (int r = ensure(h, h->jp, (size_t)B * U * JH)) return r;
  if (int r = ensure(h, h->jz, rows * JH)) return r;
  if (int r = ensure(h, h->jl, rows * V)) return r;
  // exact-fp32 MFMA GEMMs like the greedy path's projection (gemm() without split planes): enc [B T, D] and dec [B U, PH]
  GamGemmArgs ge = gemm_args(enc, D, h->jn_enc_w, h->jn_enc_b, h->encp.p, JH, B * T, JH, D);
  if (int r = gemm(h, s, ge, GAM_ACT_NONE, GAM_PF_DECODE)) return r;
  GamGemmArgs gp = gemm_args(dec, PH, h->jn_pred_w, h->jn_pred_b, h->jp.p, JH, B * U, JH, PH);
  if (int r = gemm(h, s, gp, GAM_ACT_NONE, GAM_PF_DECODE)) return r;
  {
    ProfScope ps(h, s, GAM_PF_DECODE, (double)rows * JH * 4.0);
    const int grid = (int)std::min<size_t>((rows * JH + 255) / 256, 8192);
    hipLaunchKernelGGL(gam_joint_hidden_kernel, dim3(grid), dim3(256), 0, s, h->encp.p, h->jp.p, h->jz.p, B, T, U, JH);
    HIPCHK(h, hipGetLastError());
  }
  GamGemmArgs go = gemm_args(h->jz.p, JH, h->jn_out_w, h->jn_out_b, h->jl.p, V, (int)rows, V, JH);
  if (int r = gemm(h, s, go, GAM_ACT_NONE, GAM_PF_DECODE)) return r;
  ProfScope ps(h, s, GAM_PF_DECODE, (double)rows * V * 8.0);
  hipLaunchKernelGGL(gam_log_softmax_kernel, dim3(gam_cdiv((long)rows, 4)), dim3(256), 0, s, h->jl.p, log_probs, (int)rows, V);
  HIPCHK(h, hipGetLastError());
  return 0;
}

int gam_emo_probs(gam_handle* h, const float* encoded, const int32_t* enc_len, int B, int64_t Tp, float* probs,
                  void* stream) {
  if (!h) return -1;
  if (h->cfg.head_type != GAM_HEAD_EMO || !h->has_head) return fail(h, -1, "model has no emotion head");
  if (B <= 0 || Tp <= 0 || !encoded || !probs) return fail(h, -1, "bad emotion-head arguments");
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  const int D = h->cfg.d_model, NC = h->cfg.num_classes;
  ProfScope ps(h, s, GAM_PF_DECODE, (double)B * D * Tp * 4.0);
  hipLaunchKernelGGL(gam_emo_head_kernel, dim3(B), dim3(256), (size_t)(D + NC) * sizeof(float), s, encoded, enc_len,
                     h->emo_w, h->emo_b, probs, D, (int)Tp, NC);
  HIPCHK(h, hipGetLastError());
  return 0;
}

int gam_op_gemm(gam_handle* h, const float* A, const float* W, const float* bias, float* C, int M, int N, int K, int act,
                void* stream) {
  if (!h) return -1;
  HIPCHK(h, hipSetDevice(h->device));
  GamGemmArgs g = gemm_args(A, K, W, bias, C, N, M, N, K);
  if (!split_mode(h)) return gemm(h, (hipStream_t)stream, g, act);
  // split-fp16 mode: the W planes are rebuilt on the device (unit scale) on every call -- a cache keyed
  // on the W pointer returned stale planes when an allocator handed the same address to a new matrix
  // of the same size.  This is a test / microbenchmark entry; the HIP-event profile class times only
  // the GEMM launch below, not these conversions.
  hipStream_t s = (hipStream_t)stream;
  const size_t count = ((size_t)N * K + 7) / 8 * 8;
  {
    if (int r = ensure(h, h->op_planes, count + 64)) return r;
    _Float16* hi = (_Float16*)h->op_planes.p;
    hipLaunchKernelGGL(gam_split_kernel, dim3((int)std::min<size_t>((count / 4 + 255) / 256, 4096)), dim3(256), 0, s, W, hi,
                       hi + count + 8, count / 4);
  }
  W16 w16;
  w16.hi = (_Float16*)h->op_planes.p; w16.lo = w16.hi + count + 8; w16.inv = 1.0f;
  w16.h16 = w16.hi;
  // per-row power-of-two scale of A, as the LayerNorm kernels provide it inside the encoder
  if (int r = ensure(h, h->op_rs, (size_t)M + 64)) return r;
  hipLaunchKernelGGL(gam_rowscale_kernel, dim3(gam_cdiv(M, 4)), dim3(256), 0, s, A, h->op_rs.p, M, K, (long)K);
  g.a_rs = h->op_rs.p;
  if (h->use_sp && K % 32 == 0 && N % 4 == 0 && M >= h->sp_min_m) {
    // sp32 operands by pre-passes (in the encoder the producing kernels write sp32 directly and the
    // weight planes are built at gam_finalize)
    const size_t wn = (size_t)N * K, an = (size_t)M * K;
    if (int r = ensure(h, h->op_sp, wn + 64)) return r;
    hipLaunchKernelGGL(gam_to_sp32_kernel, dim3((int)std::min<size_t>((wn / 4 + 255) / 256, 4096)), dim3(256), 0, s, W,
                       (_Float16*)h->op_sp.p, wn / 4, (const float*)nullptr, K);
    if (int r = ensure(h, h->aplanes, an + 64)) return r;
    g.a_fmt = (h->gemm_mode == GAM_GEMM_F16 && K % 64 == 0) ? 2 : 1;   // (other K: the three-term kernels, whatever the mode)
    if (g.a_fmt == 2)
      hipLaunchKernelGGL(gam_to_h16_kernel, dim3((int)std::min<size_t>((an / 4 + 255) / 256, 8192)), dim3(256), 0, s, A,
                         (_Float16*)h->aplanes.p, an / 4, (const float*)h->op_rs.p, K);
    else
      hipLaunchKernelGGL(gam_to_sp32_kernel, dim3((int)std::min<size_t>((an / 4 + 255) / 256, 8192)), dim3(256), 0, s, A,
                         (_Float16*)h->aplanes.p, an / 4, (const float*)h->op_rs.p, K);
    w16.sp = (_Float16*)h->op_sp.p;
    g.Asp = (const _Float16*)h->aplanes.p;
  } else if (K % 4 == 0) {   // 128x128 kernel: a row-scaled fp32 copy of A (what the LayerNorm kernels write inside the encoder)
    const size_t an = (size_t)M * K;
    if (int r = ensure(h, h->aplanes, an + 64)) return r;
    hipLaunchKernelGGL(gam_scale_rows_kernel, dim3((int)std::min<size_t>((an / 4 + 255) / 256, 8192)), dim3(256), 0, s, A,
                       h->aplanes.p, an / 4, (const float*)h->op_rs.p, K);
    g.A = h->aplanes.p;
  } else {
    g.a_rs = nullptr;
  }
  return gemm(h, s, g, act, GAM_PF_GEMM, &w16);
}

int gam_op_attention(gam_handle* h, const float* q, const float* k, const float* v, float* ctx, const int32_t* lens, int B,
                     int T, int H, void* stream) {
  if (!h) return -1;
  if (B <= 0 || T <= 0 || H <= 0) return fail(h, -1, "bad attention shape");
  HIPCHK(h, hipSetDevice(h->device));
  GamAttnArgs at;
  memset(&at, 0, sizeof at);
  const int D = H * GAM_ATT_DK;
  at.q = q; at.k = k; at.v = v; at.ctx = ctx; at.lens = lens;
  at.B = B; at.Ta = T; at.Tv = T; at.H = H; at.ldq = D; at.ldv = D; at.ldo = D;
  at.scale = 1.0f / sqrtf((float)GAM_ATT_DK);
  hipError_t e = gam_launch_attn_mode(at, GAM_ATT_DK, split_mode(h), (hipStream_t)stream, h->gemm_mode == GAM_GEMM_F16 ? 1 : 3, h->ncu);
  if (e != hipSuccess) return fail(h, -2, "attention launch: %s", hipGetErrorString(e));
  return 0;
}

int gam_set_gemm_mode(gam_handle* h, int mode) {
  if (!h || (mode != GAM_GEMM_F32 && mode != GAM_GEMM_F16X3 && mode != GAM_GEMM_F16)) return fail(h, -1, "unknown GEMM mode %d", mode);
  h->gemm_mode = mode;
  return 0;
}

int gam_get_gemm_mode(const gam_handle* h) { return h ? h->gemm_mode : -1; }

// Debug: FNV-1a hash of one of the decode's scratch buffers as it is in device memory now (the call synchronises the device).
// which: 0 = token-major encoder output (tok), 1 = encoder projection (encp), 2 = hand-off granules (rnnt_x), 3 = CTC logits.
int gam_debug_buffer_hash(gam_handle* h, int which, uint64_t* hash_out, int64_t* floats_out) {
  if (!h || !hash_out) return -1;
  const DevBuf* b = which == 0 ? &h->tok : which == 1 ? &h->encp : which == 2 ? &h->rnnt_x : which == 3 ? &h->logits : nullptr;
  if (!b) return fail(h, -1, "no such buffer %d", which);
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipDeviceSynchronize());
  std::vector<uint32_t> host(b->cap);
  if (b->cap) HIPCHK(h, hipMemcpy(host.data(), b->p, b->cap * sizeof(float), hipMemcpyDeviceToHost));
  uint64_t x = 1469598103934665603ull;
  for (uint32_t v : host) { x ^= v; x *= 1099511628211ull; }
  *hash_out = x;
  if (floats_out) *floats_out = (int64_t)b->cap;
  return 0;
}

int gam_set_rnnt_cluster(gam_handle* h, int workgroups_per_utterance) {
  if (!h) return -1;
  if (workgroups_per_utterance < -1 || workgroups_per_utterance > 8) return fail(h, -1, "cluster size %d outside -1 (auto) .. 8", workgroups_per_utterance);
  h->rnnt_cluster = workgroups_per_utterance;
  return 0;
}

int gam_get_rnnt_cluster(gam_handle* h) { return h ? h->rnnt_cluster : -2; }

int gam_last_encode_rows(gam_handle* h, int* rows_padded) {
  if (!h) return -1;
  if (rows_padded) *rows_padded = h->last_rows_padded;
  return h->last_rows;
}

int gam_range_flag(gam_handle* h, int* flag_host, void* stream) {
  if (!h || !flag_host) return -1;
  if (!h->finalized) return fail(h, -1, "gam_range_flag before gam_finalize");
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(h, hipMemcpyAsync(flag_host, h->range_flag, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipMemsetAsync(h->range_flag, 0, sizeof(int), s));
  HIPCHK(h, hipStreamSynchronize(s));
  return 0;
}

int gam_range_flag_fetch(gam_handle* h, int32_t* flag_dev, void* stream) {
  if (!h || !flag_dev) return -1;
  if (!h->finalized) return fail(h, -1, "gam_range_flag_fetch before gam_finalize");
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(h, hipMemcpyAsync(flag_dev, h->range_flag, sizeof(int), hipMemcpyDeviceToDevice, s));
  HIPCHK(h, hipMemsetAsync(h->range_flag, 0, sizeof(int), s));
  return 0;
}

int gam_plan_sp(int M, int N, int K, int n_cu, int* mt, int* nw, int* splitk) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 32 != 0 || !mt || !nw || !splitk) return -1;
  const GamSpPlan p = gam_gemm_sp_plan(M, N, K, 0, n_cu > 0 ? n_cu : 256);
  *mt = p.mt; *nw = p.nw; *splitk = p.s;
  return 0;
}

int gam_tune_sp(int mt, int nw, int splitk) {
  GamSpForce& f = gam_sp_force();
  f.mt = mt; f.nw = nw; f.s = splitk;
  return 0;
}

int gam_tune_sp_stages(int stages) {
  if (stages != 0 && stages != 2 && stages != 3) return -1;
  gam_sp_force().ns = stages;
  return 0;
}

int gam_plan_sp_ex(int M, int N, int K, int n_cu, int* mt, int* nw, int* splitk, int* stages) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 32 != 0 || !mt || !nw || !splitk || !stages) return -1;
  const GamSpPlan p = gam_gemm_sp_plan(M, N, K, 0, n_cu > 0 ? n_cu : 256);
  *mt = p.mt; *nw = p.nw; *splitk = p.s; *stages = p.ns;
  return 0;
}

int gam_profile_enable(gam_handle* h, int on) {
  if (!h) return -1;
  h->prof_on = on;
  h->prof_used = 0;
  for (int i = 0; i < GAM_PF_NCLASS; ++i) { h->prof_work[i] = 0; h->prof_launches[i] = 0; h->prof_bytes[i] = 0; }
  return 0;
}

int gam_profile_pause(gam_handle* h, int on) {
  if (!h) return -1;
  h->prof_on = on;      // (no reset: what was collected so far stays)
  return 0;
}

int gam_profile_read_bytes(gam_handle* h, int cls, double* bytes) {
  if (!h || cls < 0 || cls >= GAM_PF_NCLASS || !bytes) return -1;
  *bytes = h->prof_bytes[cls];
  return 0;
}

int gam_profile_read(gam_handle* h, int cls, double* ms, int64_t* launches, double* work) {
  if (!h || cls < 0 || cls >= GAM_PF_NCLASS) return -1;
  double tot = 0;
  for (size_t i = 0; i < h->prof_used; ++i) {
    ProfEvent& e = h->prof_events[i];
    if (e.cls != cls) continue;
    HIPCHK(h, hipEventSynchronize(e.b));
    float t = 0;
    HIPCHK(h, hipEventElapsedTime(&t, e.a, e.b));
    tot += t;
  }
  if (ms) *ms = tot;
  if (launches) *launches = h->prof_launches[cls];
  if (work) *work = h->prof_work[cls];
  return 0;
}

}  // extern "C"
