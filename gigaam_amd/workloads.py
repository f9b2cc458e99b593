"""Seeded synthetic workloads of the five BASELINE.json configurations, shared by bench.py,
tools/ and the tests so that the timed run, the parity checks and the golden fixtures all see the same audio.

  1  v2_ctc, one 5 s clip
  2  v2_ctc, 32 x 20 s                         (3: the same audio through v2_rnnt)
  4  v3_e2e_rnnt, 1024 utterances with durations U(5 s, 20 s), seed 1234, sorted by length into
     32-utterance batches that are dealt to the ranks (gigaam_amd/shard.py)
  5  one hour of audio: synthetic speech regions (the stand-in for pyannote's output) -> the reference's chunk
     packer (vad_utils.pack_regions: 22 s / 15 s / 30 s / 0.2 s) -> batches of 16 dealt round-robin to the ranks
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import synth

SR = 16000


def config1_clip() -> Tuple[torch.Tensor, torch.Tensor]:
    return synth.synth_audio(1, 5.0, seed=0)


def config2_batch(batch: int = 32, seconds: float = 20.0, rank: int = 0, first: Optional[int] = None):
    """Utterances ``first .. first+batch-1`` (default: rank*batch ..) of the seed-1000 stream; equal lengths."""
    i0 = rank * batch if first is None else first
    return synth.synth_audio(batch, seconds, seed=1000, index0=i0)


def config2_ragged_lengths(batch: int = 32, lo_s: float = 10.0, hi_s: float = 20.0) -> List[int]:
    """SURVEY.md §8d, config 2's second run: lengths ``linspace(10 s, 20 s, batch)`` (in samples), longest LAST -- the masks,
    the per-utterance frame counts and the padded tails are all exercised."""
    return [int(round(float(x) * SR)) for x in np.linspace(lo_s, hi_s, batch)]


def config2_ragged_batch(batch: int = 32, first: int = 0):
    """The utterances of ``config2_batch`` (same seed-1000 stream) cut to ``config2_ragged_lengths``; zero-padded to 20 s."""
    lens = config2_ragged_lengths(batch)
    return synth.synth_audio(batch, max(lens) / SR, seed=1000, index0=first, lengths=lens)


def config4_durations(n_utts: int = 1024) -> np.ndarray:
    return np.random.RandomState(1234).uniform(5.0, 20.0, size=n_utts)


def config4_batches(n_utts: int = 1024, batch: int = 32, only_batches: Optional[Sequence[int]] = None):
    """[(wav [b,L], len [b], global_indices [b])] -- batch j holds the j-th block of the length-sorted set
    (longest first), zero-padded to its own longest utterance.  ``only_batches`` generates just those."""
    from .shard import sorted_batches
    durs = config4_durations(n_utts)
    lens = (durs * SR).astype(np.int64)
    out = []
    for j, idx in enumerate(sorted_batches(lens.tolist(), batch)):
        if only_batches is not None and j not in only_batches:
            continue
        bl = [int(lens[i]) for i in idx]
        wav, wlen = synth.synth_audio(len(idx), max(bl) / SR, seed=4000 + j, lengths=bl)
        out.append((wav, wlen, list(idx)))
    return out


def config5_regions(total_seconds: int = 3600) -> List[Tuple[float, float]]:
    """2-12 s of speech separated by 0.3-1.5 s pauses (stand-in for the VAD output), seed 7."""
    rng = np.random.RandomState(7)
    regions, t = [], 0.5
    while t < total_seconds - 13:
        d = float(rng.uniform(2.0, 12.0))
        regions.append((round(t, 2), round(t + d, 2)))
        t += d + float(rng.uniform(0.3, 1.5))
    return regions


def config5_audio(total_seconds: int = 3600) -> torch.Tensor:
    """The tone/noise recipe a minute at a time, quantised like a PCM16 file (preprocess.py:40 scaling)."""
    chunks = []
    for i in range(0, total_seconds, 60):
        w, _ = synth.synth_audio(1, float(min(60, total_seconds - i)), seed=7000 + i)
        chunks.append((w[0].numpy() * 32767.0).astype(np.int16))
    return torch.from_numpy(np.concatenate(chunks).astype(np.float32) / 32768.0)


def config5_segments(total_seconds: int = 3600):
    """(segments, boundaries) exactly as transcribe_longform would cut them (vad_utils.segment_audio_file)."""
    from .vad_utils import pack_regions
    audio = config5_audio(total_seconds)
    bounds = pack_regions(config5_regions(total_seconds), audio.shape[0] / SR)
    return [audio[int(s * SR): int(e * SR)] for s, e in bounds], bounds
