"""Fixture generator (run in the build container): calibration vectors for the
synthetic checkpoints = time-average of the CPU-oracle encoder output on a fixed
seeded calibration batch.  Output: gigaam_amd/synth_calib/<key>.npy (3 KB each).

    python tests/golden/make_calib.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gigaam_amd import synth  # noqa: E402
from oracle import gigaam_oracle as O  # noqa: E402

CASES = [  # (model, seed, n_layers)
    ("v2_ctc", 0, 16), ("v3_ctc", 0, 16), ("v1_ctc", 0, 16),
    ("v2_ctc", 1, 2), ("v3_ctc", 1, 2), ("v1_ctc", 1, 2),
    ("v2_ctc", 1, 16),
]


def main():
    torch.set_num_threads(os.cpu_count())
    out_dir = os.path.join(ROOT, "gigaam_amd", "synth_calib")
    os.makedirs(out_dir, exist_ok=True)
    for name, seed, nl in CASES:
        ck = synth.make_checkpoint(name, seed=seed, calib=None, n_layers=nl)
        cfg, sd = ck["cfg"], ck["state_dict"]
        path = os.path.join(out_dir, synth.calib_key(cfg, seed) + ".npy")
        if os.path.exists(path) and "--force" not in sys.argv:
            print("exists", path)
            continue
        wav, lens = synth.synth_audio(3, 4.0, seed=1000, lengths=[64000, 50000, 33333])
        with torch.no_grad():
            feat, flen = O.log_mel(wav, lens, cfg["preprocessor"],
                                   sd["preprocessor.featurizer.0.spectrogram.window"],
                                   sd["preprocessor.featurizer.0.mel_scale.fb"])
            y, l = O.encoder_forward(sd, cfg["encoder"], feat, flen)
        m = torch.arange(y.shape[2])[None, :] < l[:, None]
        c = (y * m[:, None, :]).sum(dim=(0, 2)) / m.sum()
        np.save(path, c.numpy().astype(np.float32))
        print("wrote", path, float(c.norm()))


if __name__ == "__main__":
    main()
