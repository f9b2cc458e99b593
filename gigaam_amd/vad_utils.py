"""Longform segmentation (reference gigaam/vad_utils.py:80-136).

The reference obtains speech regions from pyannote's VoiceActivityDetection
pipeline (a gated third-party network; SURVEY.md §2.1 row 8 marks the model out
of scope) and then packs them into 15-22 s chunks with a 30 s hard cap.  Only the
packing is restated here; regions come from the caller.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from .preprocess import load_audio


def pack_regions(regions: Sequence[Tuple[float, float]], audio_seconds: float = float("inf"),
                 max_duration: float = 22.0, min_duration: float = 15.0,
                 strict_limit_duration: float = 30.0, new_chunk_threshold: float = 0.2) -> List[Tuple[float, float]]:
    """Chunk packing over (start, end) speech regions in seconds, same decisions as
    reference vad_utils.py:98-136: the open chunk [c0, c1] absorbs the next region
    unless it is already longer than ``new_chunk_threshold`` AND (absorbing would
    push it past ``max_duration`` OR it is already past ``min_duration``); a closed
    chunk longer than ``strict_limit_duration`` is cut into int(dur/limit)+1 equal
    parts; a trailing chunk shorter than ``new_chunk_threshold`` is dropped."""
    out: List[Tuple[float, float]] = []

    def close(c0: float, c1: float) -> None:
        dur = c1 - c0
        parts = int(dur / strict_limit_duration) + 1 if dur > strict_limit_duration else 1
        step = dur / parts
        lo, hi = c0, c0 + step if parts > 1 else c1
        for _ in range(parts - 1):
            out.append((lo, hi))
            lo, hi = hi, hi + step
        out.append((lo, hi))

    c0 = c1 = dur = 0.0
    for start, end in regions:
        start, end = max(0, start), min(audio_seconds, end)
        if dur == 0.0:
            c0 = start
        elif dur > new_chunk_threshold and (dur + (end - c1) > max_duration or dur > min_duration):
            close(c0, c1)
            c0 = start
        c1 = end
        dur = c1 - c0
    if dur > new_chunk_threshold:
        close(c0, c1)
    return out


def segment_audio_file(wav_file: str, sr: int, device=None,
                       speech_regions: Optional[Sequence[Tuple[float, float]]] = None,
                       vad: Optional[Callable[[Tensor, int], Sequence[Tuple[float, float]]]] = None,
                       **pack_kwargs) -> Tuple[List[Tensor], List[Tuple[float, float]]]:
    audio = load_audio(wav_file, sr)
    if speech_regions is None:
        if vad is None:
            raise RuntimeError("no VAD available: pass speech_regions=[(start,end),...] or vad=callable "
                               "(the reference's pyannote pipeline is not part of this package)")
        speech_regions = vad(audio, sr)
    bounds = pack_regions(speech_regions, audio.shape[0] / sr, **pack_kwargs)
    return [audio[int(s * sr): int(e * sr)] for s, e in bounds], bounds
