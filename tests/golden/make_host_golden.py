"""Fixtures for the host-side rows of SURVEY.md §8f (batch assembly, VAD chunk packer, word
timestamps, the longform loop's bookkeeping) -- runs ONLY in the build container.

Every value written here is the output of the REFERENCE's own, unmodified functions
(/root/reference/gigaam/{vad_utils,timestamps_utils,utils,model,decoding}.py) imported through
oracle/ref_shim.py; third-party pieces that cannot exist offline are replaced by scripted stand-ins
(the pyannote pipeline returns a scripted timeline, ffmpeg's load_audio returns a seeded tensor).

    python tests/golden/make_host_golden.py      ->  tests/golden/host_logic.json, tests/golden/spm256.model
"""
import hashlib
import importlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle.ref_shim import _stub, import_reference  # noqa: E402

SR = 16000


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()[:16]


def seeded_audio(n: int, seed: int) -> torch.Tensor:
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(g.standard_normal(n, dtype=np.float32) * np.float32(0.1))


def import_vad_utils():
    """gigaam.vad_utils with pyannote stubbed (its VAD network is a gated third-party model)."""
    pa = _stub("pyannote")
    paa = _stub("pyannote.audio", Model=object, Pipeline=object)
    pa.audio = paa
    core = _stub("pyannote.audio.core")
    task = _stub("pyannote.audio.core.task", Problem=object, Resolution=object, Specifications=object)
    core.task = task
    paa.core = core
    paa.pipelines = _stub("pyannote.audio.pipelines", VoiceActivityDetection=object)
    return importlib.import_module("gigaam.vad_utils")


class _Seg:
    def __init__(self, s, e):
        self.start, self.end = s, e


class ScriptedPipeline:
    """pipeline(wav_file).get_timeline().support() -> iterable of objects with .start/.end"""

    def __init__(self, regions):
        self.regions = regions

    def __call__(self, _wav_file):
        return self

    def get_timeline(self):
        return self

    def support(self):
        return [_Seg(s, e) for s, e in self.regions]


def random_regions(seed: int, total: float, n: int):
    g = np.random.Generator(np.random.PCG64(seed))
    cuts = np.sort(g.uniform(0.0, total, size=2 * n))
    return [(float(round(cuts[2 * i], 3)), float(round(cuts[2 * i + 1], 3))) for i in range(n)]


def vad_cases():
    cases = [
        dict(regions=[(0.0, 5.0), (5.5, 12.0), (12.5, 18.0), (18.4, 21.0), (30.0, 100.0), (100.5, 100.6)], seconds=101.0, kwargs={}),
        dict(regions=[(-0.3, 4.0), (4.1, 9.0), (9.05, 130.0)], seconds=125.5, kwargs={}),          # clipping at both ends, 4-way split
        dict(regions=[(0.0, 0.1)], seconds=3.0, kwargs={}),                                        # only a sub-threshold chunk: dropped
        dict(regions=[], seconds=3.0, kwargs={}),
        dict(regions=[(0.0, 1.1), (1.2, 2.4), (2.6, 3.3), (3.5, 5.0)], seconds=5.0, kwargs=dict(min_duration=0.5, max_duration=1.0)),
        dict(regions=[(1.0, 3.5), (4.2, 9.0), (10.5, 11.2), (13.0, 19.0)], seconds=20.0, kwargs=dict(min_duration=2.0, max_duration=6.0)),
        dict(regions=[(0.5, 29.9), (30.0, 60.2), (60.25, 61.0)], seconds=61.0, kwargs=dict(strict_limit_duration=10.0, new_chunk_threshold=1.0)),
    ]
    for seed, total, n in [(1, 600.0, 90), (2, 3600.0, 400), (3, 120.0, 7)]:
        cases.append(dict(regions=random_regions(seed, total, n), seconds=total, kwargs={}))
    cases.append(dict(regions=random_regions(4, 900.0, 150), seconds=900.0, kwargs=dict(max_duration=12.0, min_duration=8.0, strict_limit_duration=14.0)))
    return cases


def make_vad(vu):
    out = []
    for ci, c in enumerate(vad_cases()):
        n = int(c["seconds"] * SR)
        audio = seeded_audio(n, 100 + ci)
        vu.load_audio = lambda _f, audio=audio: audio
        vu.get_pipeline = lambda _d, regs=c["regions"]: ScriptedPipeline(regs)
        segs, bounds = vu.segment_audio_file("unused.wav", SR, **c["kwargs"])
        for (s, e), seg in zip(bounds, segs):
            assert torch.equal(seg, audio[int(s * SR): int(e * SR)])
        out.append(dict(regions=[list(r) for r in c["regions"]], audio_samples=n, audio_seed=100 + ci, kwargs=c["kwargs"],
                        boundaries=[[float(s), float(e)] for s, e in bounds], segment_lens=[int(s.shape[0]) for s in segs],
                        segment_sha=[sha(s) for s in segs[:4]]))
    return out


def train_spm(path_prefix: str, vocab: int = 256):
    import sentencepiece as spm
    g = np.random.Generator(np.random.PCG64(77))
    letters = list("абвгдежзийклмнопрстуфхцчшщъыьэюя")
    words = ["".join(g.choice(letters, size=int(g.integers(2, 9)))) for _ in range(4000)]
    corpus = path_prefix + "_corpus.txt"
    with open(corpus, "w", encoding="utf-8") as f:
        for _ in range(6000):
            f.write(" ".join(g.choice(words, size=int(g.integers(3, 12)))) + "\n")
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=path_prefix, vocab_size=vocab, model_type="unigram",
                                   character_coverage=1.0, input_sentence_size=6000, shuffle_input_sentence=False,
                                   num_threads=1, bos_id=-1, eos_id=-1, unk_id=0, minloglevel=2)
    os.remove(corpus)
    os.remove(path_prefix + ".vocab")
    return path_prefix + ".model"


def make_words(ref, spm_path):
    from cases import CASES, make_case_checkpoint
    ts = importlib.import_module("gigaam.timestamps_utils")
    out = []
    for case, tok_kind in [("v2_ctc_l2", "char"), ("v3_ctc_l2", "char"), ("v1_ctc_l2", "char"), ("v3_e2e_ctc_l2", "spm256"),
                           ("v2_rnnt_l2", "char")]:
        ck, wav, wlen = make_case_checkpoint(case)
        gold = dict(np.load(os.path.join(HERE, case + ".npz")))
        vocab = ck["cfg"]["decoding"]["vocabulary"]
        tok = ref.decoding.Tokenizer(vocab) if tok_kind == "char" else ref.decoding.Tokenizer([], spm_path)
        o = 0
        for i, c in enumerate(gold["counts"].tolist()):
            ids = [int(x) for x in gold["ids"][o:o + c]]
            frames = [int(x) for x in gold["frames"][o:o + c]]
            o += c
            if tok_kind == "spm256":
                ids = [x % len(tok) for x in ids]
            shift = ts.compute_frame_shift(int(wlen[i]), int(gold["enc_len"][i]))
            words = ts.frames_to_words(tok, ids, frames, shift)
            out.append(dict(case=case, utt=i, tokenizer=tok_kind, ids=ids, frames=frames, wav_len=int(wlen[i]),
                            enc_len=int(gold["enc_len"][i]), frame_shift=shift, text=tok.decode(ids),
                            words=[[w.text, w.start, w.end] for w in words]))
    # hand-made token streams: leading/trailing/double spaces, a lone marker piece
    tok = ref.decoding.Tokenizer(list(" абвгд"))
    for ids, frames in [([0, 1, 2, 0, 0, 3, 0], [0, 2, 3, 5, 6, 9, 12]), ([0, 0], [1, 2]), ([], []), ([4], [7])]:
        words = ts.frames_to_words(tok, ids, frames, 0.04)
        out.append(dict(case="hand", utt=0, tokenizer="hand:" + "".join(tok.vocab), ids=ids, frames=frames, wav_len=0, enc_len=0,
                        frame_shift=0.04, text=tok.decode(ids), words=[[w.text, w.start, w.end] for w in words]))
    return out


def make_collate(ref):
    out = []
    for seed, lens in [(1, [64000, 50000, 33333]), (2, [5]), (3, [100, 1, 77, 100]), (4, [352000, 480000, 16000, 240000, 479999])]:
        wavs = [seeded_audio(n, 1000 * seed + j) for j, n in enumerate(lens)]
        batch, lengths = ref.utils.AudioDataset.collate(wavs)
        assert batch.dtype == torch.float32 and lengths.dtype == torch.int64
        out.append(dict(seed=seed, lens=lens, shape=list(batch.shape), lengths=lengths.tolist(), batch_sha=sha(batch),
                        pad_is_zero=bool(all(float(batch[j, n:].abs().sum()) == 0.0 for j, n in enumerate(lens)))))
    return out


def scripted_decode(wav_lens, word_timestamps):
    """Stand-in for GigaAMASR._decode in the longform fixture: a deterministic function of the lengths."""
    res = []
    for n in [int(x) for x in wav_lens]:
        text = f"seg{n}"
        words = None
        if word_timestamps:
            d = n / SR
            words = [("a", 0.0104, d / 3), ("b", d / 3 + 0.0007, d * 0.9996)]
        res.append((text, words))
    return res


def make_longform(vu):
    sys.modules["hydra"].utils.instantiate = lambda *a, **k: None
    sys.modules["torchaudio"].load = None
    rm = importlib.import_module("gigaam.model")
    out = []
    for ci, (regions, seconds, bs, wts) in enumerate([
            (random_regions(11, 300.0, 40), 300.0, 4, True), (random_regions(12, 200.0, 25), 200.0, 16, False),
            ([(0.0, 0.05)], 2.0, 16, True)]):
        n = int(seconds * SR)
        audio = seeded_audio(n, 500 + ci)
        vu.load_audio = lambda _f, audio=audio: audio
        vu.get_pipeline = lambda _d, regs=regions: ScriptedPipeline(regs)
        seen = []

        def fwd(wav_pad, wav_lens):
            seen.append(dict(shape=list(wav_pad.shape), lengths=wav_lens.tolist(), sha=sha(wav_pad)))
            return wav_pad, wav_lens

        def dec(encoded, encoded_len, wav_lens, word_timestamps):
            return [(t, None if w is None else [rm.Word(text=a, start=s, end=e) for a, s, e in w])
                    for t, w in scripted_decode(wav_lens.tolist(), word_timestamps)]

        fake = types.SimpleNamespace(_device=torch.device("cpu"), _dtype=torch.float32, forward=fwd, _decode=dec)
        res = rm.GigaAMASR.transcribe_longform.__wrapped__(fake, "unused.wav", word_timestamps=wts, fr_batch_size=bs) \
            if hasattr(rm.GigaAMASR.transcribe_longform, "__wrapped__") else \
            rm.GigaAMASR.transcribe_longform(fake, "unused.wav", word_timestamps=wts, fr_batch_size=bs)
        out.append(dict(regions=[list(r) for r in regions], audio_samples=n, audio_seed=500 + ci, fr_batch_size=bs,
                        word_timestamps=wts, batches=seen,
                        segments=[dict(text=s.text, start=s.start, end=s.end,
                                       words=None if s.words is None else [[w.text, w.start, w.end] for w in s.words])
                                  for s in res.segments]))
    return out


def main():
    ref = import_reference()
    vu = import_vad_utils()
    spm_path = train_spm(os.path.join(HERE, "spm256"))
    fx = dict(vad_pack=make_vad(vu), words=make_words(ref, spm_path), collate=make_collate(ref), longform=make_longform(vu),
              note="outputs of the reference's own functions; see tests/golden/make_host_golden.py")
    with open(os.path.join(HERE, "host_logic.json"), "w", encoding="utf-8") as f:
        json.dump(fx, f, ensure_ascii=False, indent=0)
    print({k: (len(v) if isinstance(v, list) else v) for k, v in fx.items()})


if __name__ == "__main__":
    main()
