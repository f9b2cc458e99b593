"""Shared helpers of the parity tests: golden cases and margin-aware comparison."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from gigaam_amd import synth  # noqa: E402
from oracle import gigaam_oracle as O  # noqa: E402

# must match tests/golden/make_golden.py
CASES = {
    "v2_ctc_l2": ("v2_ctc", 1, 2, (3, 4.0, 11, [64000, 50000, 33333])),
    "v2_ctc_l2_b1": ("v2_ctc", 1, 2, (1, 2.5, 12, None)),
    "v2_rnnt_l2": ("v2_rnnt", 1, 2, (3, 4.0, 13, [64000, 41234, 57000])),
    "v3_ctc_l2": ("v3_ctc", 1, 2, (3, 4.0, 14, [64000, 50000, 33333])),
    "v3_e2e_rnnt_l2": ("v3_e2e_rnnt", 1, 2, (2, 3.0, 15, [48000, 30011])),
    "v1_ctc_l2": ("v1_ctc", 1, 2, (3, 3.0, 16, [48000, 40000, 20000])),
    "v2_ctc_l2_short": ("v2_ctc", 1, 2, (2, 0.3125, 17, [5000, 3200])),
    "v3_e2e_ctc_l2": ("v3_e2e_ctc", 1, 2, (2, 3.0, 19, [48000, 35000])),
    "v1_rnnt_l2": ("v1_rnnt", 1, 2, (2, 2.5, 20, [40000, 26000])),
}
EMO_CASE = ("emo", 1, 2, (2, 3.0, 18, [48000, 36000]))   # tests/golden/emo_l2.npz

# tolerances (fp32 vs fp32, different summation orders)
TOL_FEAT = 2e-3      # log-mel, natural-log units
# Mel bands more than 60 dB below their frame's strongest band are differences of large DFT terms: there an
# fp32 DFT-by-matmul (this path) and an fp32 FFT (torch.stft, the oracle and the reference) legitimately
# disagree by a few 1e-3 in the log (both are ~1e-6 relative to the frame norm); the bar there is looser.
TOL_FEAT_WEAK = 2e-2
WEAK_BAND_DB = 60.0


def logmel_err(feat, feat_ref, frame_mask=None):
    """(max abs error over bands within WEAK_BAND_DB of their frame's peak, max abs error over the weaker
    ones); feat [B,M,T] natural-log mel power; frame_mask [B,T] selects valid frames."""
    d = (feat - feat_ref).abs()
    strong = feat_ref >= feat_ref.max(dim=1, keepdim=True).values - WEAK_BAND_DB * 0.2302585
    if frame_mask is not None:
        d = d * frame_mask[:, None, :]
    return float((d * strong).max()), float((d * ~strong).max())
TOL_ENC = 2e-4       # encoder activations (LayerNorm-scaled, O(1) values)
# the stem output is not LayerNorm-scaled (|values| up to ~6 in the goldens) and is a sum of ~7000-12000 products of
# large post-ReLU activations that cancel: its bar scales with the tensor's magnitude
def tol_pre(ref):
    return 1e-4 + 5e-5 * float(ref.abs().max())


TOL_LOGP = 1e-3      # CTC / RNN-T log-probs: BASELINE.json north_star "within 1e-3 fp32"


def load_case(name):
    model, seed, nl, (b, secs, aseed, lens) = CASES[name]
    ck = synth.make_checkpoint(model, seed=seed, n_layers=nl)
    wav, wlen = synth.synth_audio(b, secs, seed=aseed, lengths=lens)
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")))
    return ck, wav, wlen, gold


def oracle_features(ck, wav, wlen):
    sd = ck["state_dict"]
    with torch.no_grad():
        return O.log_mel(wav, wlen, ck["cfg"]["preprocessor"], sd["preprocessor.featurizer.0.spectrogram.window"],
                         sd["preprocessor.featurizer.0.mel_scale.fb"])


def valid_mask(t, lens):
    return torch.arange(t)[None, :] < torch.as_tensor(lens)[:, None].to(torch.int64)


def split_ragged(flat_ids, flat_frames, counts):
    out, o = [], 0
    for c in counts:
        out.append((list(map(int, flat_ids[o:o + c])), list(map(int, flat_frames[o:o + c]))))
        o += c
    return out


def ragged_from_device(ids, frames, counts):
    n = counts.cpu().tolist()
    return [(ids[i, :c].cpu().tolist(), frames[i, :c].cpu().tolist()) for i, c in enumerate(n)]
