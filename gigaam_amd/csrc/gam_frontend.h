// gam_frontend.h -- log-mel frontend (reference gigaam/preprocess.py:43-98 wrapping
// torchaudio.transforms.MelSpectrogram; contract restated in SURVEY.md §8c):
//   reflect-pad n_fft/2 (center=True) -> frames of n_fft samples every hop -> periodic
//   Hann -> rFFT -> |X|^2 -> mel = fb^T . power (HTK, norm=None) -> log(clamp(.,1e-9,1e9)).
// The windowed DFT is one fp32 MFMA GEMM (gam_gemm.h) whose A operand is the padded
// waveform itself read with overlapping rows (lda = hop) and whose W operand is the
// window-folded cos/sin basis built once at finalize; only two small HBM-bound kernels
// remain: the padding producer and the power/mel/log consumer.
#pragma once
#include "gam_common.h"

// wav [B, L] -> wavp [B, Lp] (+slack);  center: wavp[i] = reflect(wav, i - half)
__global__ __launch_bounds__(256) void gam_pad_wav_kernel(const float* wav, float* wavp, int B, long L, long Lp,
                                                          int half, int center) {
  const int b = blockIdx.y;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= Lp) return;
  float v = 0.f;
  if (center) {
    long j = i - half;
    if (j < 0) j = -j;
    else if (j >= L) j = 2 * (L - 1) - j;
    if (i < L + 2 * half && j >= 0 && j < L) v = wav[(size_t)b * L + j];
  } else {
    if (i < L) v = wav[(size_t)b * L + i];
  }
  wavp[(size_t)b * Lp + i] = v;
}

struct GamPowMelArgs {
  const float* spec;   // [B*Tfa, lds]: re[0..nf) | im[nf..2nf)
  const float* wts;    // the non-zero stretch of each band's filter (column of the [nf, n_mels] filterbank), band after band
  int nw;              // floats in wts
  const int* band;     // [3 * n_mels] per mel band: first / one-past-last bin with a non-zero weight, offset into wts
  float* feat;         // [B, n_mels, Tf]
  const long long* wav_len;  // [B] samples (may be null -> no feat_len output)
  long long* feat_len;       // [B]
  int B, Tfa, Tf, nf, n_mels, lds;
  int hop, win, center;
};

// block: 64 frames x (n_mels <= 64) ; thread = (frame, 16 mel bands)
__global__ __launch_bounds__(256) void gam_powmel_kernel(GamPowMelArgs a) {
  extern __shared__ float gam_smem_pm[];   // power [64][nf + 1] | filter weights [nw]
  const int tid = threadIdx.x;
  const int b = blockIdx.y, t0 = blockIdx.x * 64;
  const int pld = a.nf + 1;
  if (blockIdx.x == 0 && tid == 0 && a.wav_len != nullptr) {
    const long long l = a.wav_len[b];
    long long num = a.center ? l : l - a.win;
    // floor division (torch div rounding_mode="floor"), preprocess.py:82-92
    long long qd = num / a.hop;
    if ((num % a.hop != 0) && (num < 0)) qd -= 1;
    a.feat_len[b] = qd + 1;
  }
  // power spectrum of the block's 64 frames: wave w takes frames w, w + 4, ..; a frame's bins are read lane-contiguous
  // (no per-element index division), FR frames = 8 FR loads per lane in flight before the first use -- at small batches the
  // kernel is a chain of L2 / HBM round trips, one per group of frames
  float* wl = gam_smem_pm + 64 * pld;   // the filter weights, staged once per block
  for (int i = tid; i < a.nw; i += 256) wl[i] = a.wts[i];
  {
    const int lane = tid & 63, wv = tid >> 6;
    constexpr int FR = 4;
    for (int g = 0; g < 16; g += FR) {
      float re[FR][4], im[FR][4];
#pragma unroll
      for (int q = 0; q < FR; ++q) {
        const int fl = wv + 4 * (g + q);
        int t = t0 + fl;
        t = t < a.Tf ? t : a.Tf - 1;
        const float* sp = a.spec + ((size_t)b * a.Tfa + t) * a.lds;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int f = lane + 64 * u, fc = f < a.nf ? f : a.nf - 1;
          re[q][u] = sp[fc];
          im[q][u] = sp[a.nf + fc];
        }
      }
#pragma unroll
      for (int q = 0; q < FR; ++q) {
        const int fl = wv + 4 * (g + q);
        const bool in = t0 + fl < a.Tf;
        float* pr = gam_smem_pm + fl * pld;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int f = lane + 64 * u;
          if (f < a.nf) pr[f] = in ? re[q][u] * re[q][u] + im[q][u] * im[q][u] : 0.f;
        }
        if (a.nf > 256) {   // (n_fft > 510: not a published configuration)
          const float* sp = a.spec + ((size_t)b * a.Tfa + (in ? t0 + fl : a.Tf - 1)) * a.lds;
          for (int f = lane + 256; f < a.nf; f += 64) {
            const float r2 = sp[f], i2 = sp[a.nf + f];
            pr[f] = in ? r2 * r2 + i2 * i2 : 0.f;
          }
        }
      }
    }
  }
  __syncthreads();
  // mel projection: the filterbank is triangular (a bin feeds <= 2 bands), so each band sums only its
  // own bins, in increasing bin order like the dense product (zero terms dropped).  Bands are dealt
  // round-robin to the 4 waves (m = 4 j + wave) so the wide high-frequency bands spread evenly; both operands come from
  // LDS (the weight address is wave-uniform: a broadcast read).
  const int fl = tid & 63, mg = tid >> 6;
  float acc[16];
  const float* pw = gam_smem_pm + fl * pld;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int m = 4 * j + mg;
    float s = 0.f;
    if (m < a.n_mels) {
      const int f0 = a.band[3 * m], f1 = a.band[3 * m + 1];   // wave-uniform
      const float* wm = wl + a.band[3 * m + 2] - f0;
      for (int f = f0; f < f1; ++f) s = fmaf(pw[f], wm[f], s);
    }
    acc[j] = s;
  }
  const int t = t0 + fl;
  if (t < a.Tf) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int m = 4 * j + mg;
      if (m < a.n_mels) {
        float v = acc[j];
        v = fminf(fmaxf(v, 1e-9f), 1e9f);
        a.feat[((size_t)b * a.n_mels + m) * a.Tf + t] = logf(v);
      }
    }
  }
}
