"""Pin row a1 (log-mel frontend) to the reference -- runs wherever `torchaudio` is importable.

torchaudio is neither installed in the build container nor vendored under /root/reference (the reference pins
it only as `torchaudio>=2.6`, pyproject.toml:30-33), so a1 is "parity unpinned" until someone runs THIS script
on a box that has torchaudio.  It computes, for every golden case, exactly what reference
gigaam/preprocess.py:60-76,94-98 computes -- torchaudio.transforms.MelSpectrogram(sample_rate, n_mels,
win_length, hop_length, n_fft, center) followed by log(clamp(x, 1e-9, 1e9)) -- on the seeded audio of the
case, and writes tests/golden/frontend_<case>.npz (feat, feat_len).  The reference's own FeatureExtractor
class is used when /root/reference (or $GIGAAM_REFERENCE_ROOT) is present; otherwise the two torchaudio calls
are made directly with the same arguments.

tests/test_oracle_golden.py::test_frontend_against_torchaudio_fixture and the -m gpu frontend test consume the
fixtures when they exist and say "parity unpinned" when they do not.

    python tests/golden/make_frontend_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def reference_featurizer(pre_cfg):
    import torchaudio
    kw = {k: v for k, v in pre_cfg.items() if k not in ("_target_", "sample_rate", "features")}
    ref_root = os.environ.get("GIGAAM_REFERENCE_ROOT", "/root/reference")
    if os.path.isdir(os.path.join(ref_root, "gigaam")):
        import importlib.util
        spec = importlib.util.spec_from_file_location("_ref_preprocess", os.path.join(ref_root, "gigaam", "preprocess.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod.FeatureExtractor(pre_cfg["sample_rate"], pre_cfg["features"], **kw).eval()
    sr = pre_cfg["sample_rate"]
    hop, win = kw.get("hop_length", sr // 100), kw.get("win_length", sr // 40)
    n_fft, center = kw.get("n_fft", sr // 40), kw.get("center", True)
    mel = torchaudio.transforms.MelSpectrogram(sample_rate=sr, n_mels=pre_cfg["features"], win_length=win, hop_length=hop,
                                               n_fft=n_fft, center=center)

    def fwd(x, length):
        out_len = (length.div(hop, rounding_mode="floor").add(1) if center
                   else (length - win).div(hop, rounding_mode="floor").add(1)).long()
        return torch.log(mel(x).clamp_(1e-9, 1e9)), out_len
    return fwd


def main():
    try:
        import torchaudio  # noqa: F401
    except ImportError:
        print("torchaudio is not importable here: row a1 stays parity-unpinned (nothing written)")
        return 2
    from cases import CASES, make_case_checkpoint
    done = set()
    for case in CASES:
        ck, wav, wlen = make_case_checkpoint(case)
        pre = ck["cfg"]["preprocessor"]
        fe = reference_featurizer(pre)
        with torch.no_grad():
            feat, flen = fe(wav.clone(), wlen)
        np.savez_compressed(os.path.join(HERE, f"frontend_{case}.npz"), feat=feat.numpy(), feat_len=flen.numpy(),
                            torchaudio_version=str(torchaudio.__version__))
        done.add(case)
        print("wrote", case, tuple(feat.shape))
    return 0


if __name__ == "__main__":
    sys.exit(main())
