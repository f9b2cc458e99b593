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
}
EMO_CASE = ("emo", 1, 2, (2, 3.0, 18, [48000, 36000]))   # tests/golden/emo_l2.npz

# tolerances (fp32 vs fp32, different summation orders)
TOL_FEAT = 2e-3      # log-mel, natural-log units (small-power bins amplify round-off)
TOL_ENC = 2e-4       # encoder activations (LayerNorm-scaled, O(1) values)
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
