"""CPU: the host-side rows of SURVEY.md §8f against fixtures written by the REFERENCE's own functions
(tests/golden/make_host_golden.py -> tests/golden/host_logic.json): VAD chunk packer
(reference gigaam/vad_utils.py:98-136), word timestamps (timestamps_utils.py:8-53), batch assembly
(utils.py:371-380) and the longform loop's bookkeeping (model.py:219-258)."""
import hashlib
import json
import os
import warnings

import numpy as np
import pytest
import torch

from common import ROOT

import gigaam_amd
from gigaam_amd import synth

GOLD = os.path.join(ROOT, "tests", "golden")
FX = json.load(open(os.path.join(GOLD, "host_logic.json"), encoding="utf-8"))
SR = 16000


def sha(t):
    return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()[:16]


def seeded_audio(n, seed):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(g.standard_normal(n, dtype=np.float32) * np.float32(0.1))


@pytest.mark.parametrize("i", range(len(FX["vad_pack"])))
def test_vad_packer_matches_reference(i, monkeypatch):
    from gigaam_amd import vad_utils
    c = FX["vad_pack"][i]
    regions = [tuple(r) for r in c["regions"]]
    got = vad_utils.pack_regions(regions, c["audio_samples"] / SR, **c["kwargs"])
    assert [list(b) for b in got] == c["boundaries"]          # exact float equality: same arithmetic, same order
    audio = seeded_audio(c["audio_samples"], c["audio_seed"])
    monkeypatch.setattr(vad_utils, "load_audio", lambda _f, _sr=SR: audio)
    segs, bounds = vad_utils.segment_audio_file("unused.wav", SR, speech_regions=regions, **c["kwargs"])
    assert [list(b) for b in bounds] == c["boundaries"] and [int(s.shape[0]) for s in segs] == c["segment_lens"]
    assert [sha(s) for s in segs[:4]] == c["segment_sha"]


def _tokenizer(kind):
    from gigaam_amd.decoding import Tokenizer
    if kind == "char":
        return Tokenizer(synth.CHAR_VOCAB)
    if kind == "spm256":
        return Tokenizer([], os.path.join(GOLD, "spm256.model"))
    assert kind.startswith("hand:")
    return Tokenizer(list(kind[5:]))


@pytest.mark.parametrize("i", range(len(FX["words"])))
def test_word_timestamps_match_reference(i):
    from gigaam_amd.timestamps_utils import compute_frame_shift, frames_to_words
    c = FX["words"][i]
    tok = _tokenizer(c["tokenizer"])
    shift = c["frame_shift"]
    if c["enc_len"]:
        assert compute_frame_shift(c["wav_len"], c["enc_len"]) == shift
    assert tok.decode(c["ids"]) == c["text"]
    words = frames_to_words(tok, c["ids"], c["frames"], shift)
    assert [[w.text, w.start, w.end] for w in words] == c["words"]


@pytest.mark.parametrize("i", range(len(FX["collate"])))
def test_collate_layout_matches_reference(i):
    from gigaam_amd.feeder import collate
    c = FX["collate"][i]
    wavs = [seeded_audio(n, 1000 * c["seed"] + j) for j, n in enumerate(c["lens"])]
    batch, lengths = collate(wavs)
    assert list(batch.shape) == c["shape"] and lengths.dtype == torch.int64 and lengths.tolist() == c["lengths"]
    assert batch.dtype == torch.float32 and sha(batch) == c["batch_sha"]
    # in-place variant on a (dirty) staging buffer, as the pinned feeder uses it
    stage = torch.full((batch.numel() + 7,), 3.0)
    b2, l2 = collate(wavs, out=stage)
    assert torch.equal(b2, batch) and torch.equal(l2, lengths)


class _CpuFeeder:
    """BatchFeeder stand-in for a box without a GPU: same collate, no pinned staging / side stream."""

    def __init__(self, segments, batch_size, device):
        self.segments, self.batch_size = segments, batch_size

    def __iter__(self):
        from gigaam_amd.feeder import batches, collate
        for chunk in batches(self.segments, self.batch_size):
            yield collate(chunk)


@pytest.mark.parametrize("i", range(len(FX["longform"])))
def test_longform_loop_matches_reference(i, monkeypatch):
    """transcribe_longform's bookkeeping (segment <-> boundary pairing, batch layout, word offsets rounded to
    3 decimals) with the model's compute replaced by the same scripted decode the fixture generator used."""
    from gigaam_amd import feeder, vad_utils
    from gigaam_amd.types import Word
    c = FX["longform"][i]
    audio = seeded_audio(c["audio_samples"], c["audio_seed"])
    monkeypatch.setattr(vad_utils, "load_audio", lambda _f, _sr=SR: audio)
    monkeypatch.setattr(feeder, "BatchFeeder", _CpuFeeder)
    model = gigaam_amd.model_from_checkpoint(synth.make_checkpoint("v2_ctc", seed=1, n_layers=1), "cpu")
    seen = []

    def scripted(wav, lens, word_timestamps=False):
        seen.append(dict(shape=list(wav.shape), lengths=lens.tolist(), sha=sha(wav)))
        out = []
        for n in lens.tolist():
            d = n / SR
            words = [Word("a", 0.0104, d / 3), Word("b", d / 3 + 0.0007, d * 0.9996)] if word_timestamps else None
            out.append((f"seg{n}", words))
        return out

    # the loop's public seam: launch_batch (device half, no sync) / collect_batch (host half)
    monkeypatch.setattr(model, "launch_batch", lambda wav, lens: (wav, lens))
    monkeypatch.setattr(model, "collect_batch", lambda h, word_timestamps=False: scripted(h[0], h[1], word_timestamps))
    res = model.transcribe_longform("unused.wav", word_timestamps=c["word_timestamps"], fr_batch_size=c["fr_batch_size"],
                                    speech_regions=[tuple(r) for r in c["regions"]])
    assert seen == c["batches"]
    got = [dict(text=s.text, start=s.start, end=s.end, words=None if s.words is None else [[w.text, w.start, w.end] for w in s.words])
           for s in res.segments]
    assert got == c["segments"]


def test_longform_default_vad_falls_back_loudly(monkeypatch):
    """transcribe_longform(wav) with no VAD argument must behave like the reference's signature
    (model.py:195-259): pyannote when importable, else the labelled EnergyVAD stand-in with a warning -- never
    an exception about missing arguments."""
    from gigaam_amd import feeder, vad_utils
    try:
        import pyannote.audio  # noqa: F401
        pytest.skip("pyannote installed: the default path is the reference's")
    except ImportError:
        pass
    audio = seeded_audio(3 * SR, 1)
    monkeypatch.setattr(vad_utils, "load_audio", lambda _f, _sr=SR: audio)
    monkeypatch.setattr(feeder, "BatchFeeder", _CpuFeeder)
    monkeypatch.setattr(vad_utils.EnergyVAD, "__call__", lambda self, a, sr: [(0.0, 1.0), (1.5, 3.0)])
    model = gigaam_amd.model_from_checkpoint(synth.make_checkpoint("v2_ctc", seed=1, n_layers=1), "cpu")
    monkeypatch.setattr(model, "launch_batch", lambda wav, lens: (wav, lens))
    monkeypatch.setattr(model, "collect_batch", lambda h, word_timestamps=False: [("x", None)] * h[0].shape[0])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        res = model.transcribe_longform("unused.wav", min_duration=0.5, max_duration=1.2)
    assert any("EnergyVAD" in str(x.message) for x in w)
    assert [(s.start, s.end) for s in res.segments] == [(0.0, 1.0), (1.5, 3.0)]
