"""Longform segmentation (reference gigaam/vad_utils.py:80-136).

The reference obtains speech regions from pyannote's VoiceActivityDetection
pipeline (a gated third-party network; SURVEY.md §2.1 row 8 marks the model out
of scope) and then packs them into 15-22 s chunks with a 30 s hard cap.  Only the
packing is restated here; regions come from the caller, or from ``EnergyVAD`` -- a
log-mel-energy detector on the HIP frontend that is a STAND-IN, NOT the reference's
pyannote model (different decisions on real speech; it only makes the longform path
runnable end to end without the gated network).
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
        if vad is None:   # the reference's default: pyannote's VAD pipeline on the file (vad_utils.py:100-101)
            speech_regions = pyannote_regions(wav_file, device)
        else:
            speech_regions = vad(audio, sr)
    bounds = pack_regions(speech_regions, audio.shape[0] / sr, **pack_kwargs)
    return [audio[int(s * sr): int(e * sr)] for s, e in bounds], bounds


_PIPELINE = None


def pyannote_regions(wav_file: str, device=None, model_id: str = "pyannote/segmentation-3.0") -> List[Tuple[float, float]]:
    """The reference's segmentation source (gigaam/vad_utils.py:17-77,100-101): pyannote's
    VoiceActivityDetection pipeline over ``pyannote/segmentation-3.0`` (local snapshot, else HF_TOKEN
    download), ``min_duration_on = min_duration_off = 0``, loaded once.  pyannote is an optional third-party
    dependency (the reference lists it under its ``longform`` extra); when it is not importable this raises
    ImportError and the caller decides (GigaAMASR.transcribe_longform falls back to EnergyVAD, loudly)."""
    global _PIPELINE
    import os

    from huggingface_hub import snapshot_download
    from pyannote.audio import Model
    from pyannote.audio.core.task import Problem, Resolution, Specifications
    from pyannote.audio.pipelines import VoiceActivityDetection
    from torch.torch_version import TorchVersion

    if _PIPELINE is None:
        try:
            local = snapshot_download(repo_id=model_id, local_files_only=True)
        except Exception:
            token = os.getenv("HF_TOKEN")
            if not token:
                raise RuntimeError(f"Model {model_id} was not found locally, and no HF_TOKEN was provided to download it.")
            local = snapshot_download(repo_id=model_id, token=token)
        with torch.serialization.safe_globals([TorchVersion, Problem, Specifications, Resolution]):
            seg_model = Model.from_pretrained(local)
        _PIPELINE = VoiceActivityDetection(segmentation=seg_model)
        _PIPELINE.instantiate({"min_duration_on": 0.0, "min_duration_off": 0.0})
    pipe = _PIPELINE.to(torch.device(device) if device is not None else torch.device("cpu"))
    return [(float(s.start), float(s.end)) for s in pipe(wav_file).get_timeline().support()]


class EnergyVAD:
    """Energy-based voice activity detector -- a stand-in for the reference's pyannote pipeline
    (reference gigaam/vad_utils.py:41-77), NOT a reimplementation of it.

    Frame energies are the mean log-mel of the model's own HIP frontend (10 ms hop, computed on
    the GPU a minute at a time); the decision logic is a few lines of host code: a threshold
    between the 10th and 95th percentile of the file's frame energies, then the same kind of
    post-processing pyannote's pipeline applies (drop speech shorter than ``min_duration_on``,
    fill pauses shorter than ``min_duration_off``)."""

    def __init__(self, preprocessor, threshold: float = 0.35, min_duration_on: float = 0.25,
                 min_duration_off: float = 0.25, pad: float = 0.05, floor_db: float = 6.0):
        self.pre = preprocessor
        self.threshold, self.min_on, self.min_off, self.pad, self.floor_db = threshold, min_duration_on, min_duration_off, pad, floor_db

    @torch.inference_mode()
    def frame_energy(self, audio: Tensor, sr: int) -> Tensor:
        dev = self.pre.engine.device
        hop = self.pre.hop_length
        step = 60 * sr // hop * hop              # whole frames per piece
        out = []
        for i in range(0, audio.shape[0], step):
            piece = audio[i:i + step]
            if piece.shape[0] < self.pre.win_length:
                break
            feat, n = self.pre(piece[None].to(dev, torch.float32), torch.tensor([piece.shape[0]], device=dev))
            k = min(int(n[0]), piece.shape[0] // hop)        # frames that start inside this piece
            out.append(feat[0, :, :k].mean(dim=0).cpu())
        return torch.cat(out) if out else torch.zeros(0)

    def __call__(self, audio: Tensor, sr: int) -> List[Tuple[float, float]]:
        e = self.frame_energy(audio, sr)
        if e.numel() == 0:
            return []
        hop_s = self.pre.hop_length / sr
        lo, hi = torch.quantile(e, 0.10).item(), torch.quantile(e, 0.95).item()
        if hi - lo < self.floor_db * 0.2303:      # natural-log units: no dynamic range -> one region
            return [(0.0, audio.shape[0] / sr)]
        active = (e > lo + self.threshold * (hi - lo)).tolist()
        runs, start = [], None
        for i, a in enumerate(active + [False]):
            if a and start is None:
                start = i
            elif not a and start is not None:
                runs.append([start * hop_s, i * hop_s])
                start = None
        merged: List[List[float]] = []
        for r in runs:                            # fill short pauses, then drop short speech
            if merged and r[0] - merged[-1][1] < self.min_off:
                merged[-1][1] = r[1]
            else:
                merged.append(r)
        total = audio.shape[0] / sr
        return [(max(0.0, a - self.pad), min(total, b + self.pad)) for a, b in merged if b - a >= self.min_on]
