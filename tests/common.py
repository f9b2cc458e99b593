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

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from cases import CASES, EMO_CASE, RNNT_MIN_MARGIN, make_case_checkpoint  # noqa: E402  (shared with make_golden.py)

# tolerances (fp32 vs fp32, different summation orders)
TOL_FEAT = 5e-4      # log-mel, natural-log units (measured on the GPU, r02: 1.5e-4)
# Mel bands more than 60 dB below their frame's strongest band are differences of large DFT terms: there an
# fp32 DFT-by-matmul (this path) and an fp32 FFT (torch.stft, the oracle and the reference) legitimately
# disagree by a few 1e-3 in the log (both are ~1e-6 relative to the frame norm); the bar there is looser.
TOL_FEAT_WEAK = 1e-2   # (measured: 3.7e-3)
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


def report(name, **values):
    """Append measured errors to $GAM_TEST_REPORT (a .jsonl) so tolerances can be read beside what was measured."""
    path = os.environ.get("GAM_TEST_REPORT")
    if path:
        import json
        with open(path, "a") as f:
            f.write(json.dumps(dict(test=name, **values)) + "\n")


def frontend_fixture(name):
    """torchaudio-made log-mel of a case (tests/golden/make_frontend_golden.py), or None: the build container has no
    torchaudio, so until someone runs that script elsewhere row a1 is parity-unpinned."""
    path = os.path.join(ROOT, "tests", "golden", f"frontend_{name}.npz")
    return dict(np.load(path)) if os.path.exists(path) else None


def load_case(name):
    ck, wav, wlen = make_case_checkpoint(name)
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")))
    return ck, wav, wlen, gold


def golden_trace(gold):
    """Per-utterance joint log-probs of the REFERENCE's RNN-T greedy decode, in decode order."""
    out, o = [], 0
    for c in gold["trace_counts"].tolist():
        out.append(torch.from_numpy(gold["trace"][o:o + c]))
        o += c
    return out


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
