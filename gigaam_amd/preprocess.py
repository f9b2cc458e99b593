"""``cfg.preprocessor`` slot: FeatureExtractor on the HIP frontend.

Mirrors reference gigaam/preprocess.py:12-98 (same constructor kwargs, call
contract ``(wav [B,L], len [B]) -> (feat [B,n_mels,T], len i64 [B])`` and
``out_len``); the compute is gam_frontend in libgigaam_hip.so.
"""
from __future__ import annotations

import shutil
import subprocess
import wave
from typing import Any, Dict, Mapping, Optional, Tuple

import numpy as np
import torch
from torch import Tensor, nn

from .engine import HipEngine, build_config

SAMPLE_RATE = 16000


def load_audio(audio_path: str, sample_rate: int = SAMPLE_RATE) -> Tensor:
    """Decode to mono ``sample_rate`` PCM16 and scale by 1/32768 (reference
    preprocess.py:12-40 pipes the file through ffmpeg).  Where no ffmpeg binary
    exists (this image), a PCM16 mono WAV at the right rate is read with the
    stdlib ``wave`` module -- identical samples, no resampling -- and anything
    else raises the reference's ``RuntimeError("Failed to load audio")``."""
    if shutil.which("ffmpeg") is not None:
        cmd = ["ffmpeg", "-nostdin", "-threads", "0", "-i", audio_path, "-f", "s16le", "-ac", "1",
               "-acodec", "pcm_s16le", "-ar", str(sample_rate), "-"]
        try:
            raw = subprocess.run(cmd, capture_output=True, check=True).stdout
        except subprocess.CalledProcessError as exc:
            raise RuntimeError("Failed to load audio") from exc
        pcm = np.frombuffer(raw, dtype=np.int16)
    else:
        try:
            with wave.open(audio_path, "rb") as wf:
                ok = wf.getnchannels() == 1 and wf.getsampwidth() == 2 and wf.getframerate() == sample_rate
                if not ok:
                    raise RuntimeError("ffmpeg is not installed and the file is not mono PCM16 at the target rate")
                pcm = np.frombuffer(wf.readframes(wf.getnframes()), dtype=np.int16)
        except (wave.Error, OSError, EOFError) as exc:
            raise RuntimeError("Failed to load audio") from exc
    return torch.from_numpy(pcm.astype(np.float32)) / 32768.0


class _EngineModule(nn.Module):
    """Base of the operator shims: holds the cfg of its slot and a (possibly shared)
    HipEngine.  ``_anchor`` is the one real parameter the reference's
    ``_device``/``_dtype`` properties need (gigaam/model.py:39-45)."""

    _prefix = ""

    def __init__(self) -> None:
        super().__init__()
        self._anchor = nn.Parameter(torch.zeros(1), requires_grad=False)
        self._engine: Optional[HipEngine] = None
        self._pending: Dict[str, Tensor] = {}

    # weights live inside the library, not in torch parameters
    def load_state_dict(self, state_dict: Mapping[str, Tensor], strict: bool = True, assign: bool = False):  # type: ignore[override]
        self._pending = {self._prefix + k: v for k, v in state_dict.items() if isinstance(v, Tensor)}
        self._engine = None
        return nn.modules.module._IncompatibleKeys([], [])

    def attach(self, engine: HipEngine) -> None:
        self._engine = engine
        self._pending = {}

    def _cfg_trees(self) -> Tuple[Any, Any, Any]:
        raise NotImplementedError

    @property
    def engine(self) -> HipEngine:
        if self._engine is None:
            pre, enc, head = self._cfg_trees()
            self._engine = HipEngine(build_config(pre, enc, head), self._pending, self._anchor.device)
        return self._engine

    def half(self):
        """``model.encoder.half()`` (reference gigaam/__init__.py:188-189, fp16_encoder=True on a GPU): the storage
        contract follows -- ``_dtype`` becomes float16 (the anchor parameter is converted like any other), inputs
        arrive as float16 and ``encoded`` leaves as float16 -- while the kernels keep computing in fp32."""
        return super().half()


class FeatureExtractor(_EngineModule):
    _prefix = "preprocessor."

    def __init__(self, sample_rate: int, features: int, **kwargs: Any):
        super().__init__()
        self.sample_rate = sample_rate
        self.features = features
        self.hop_length = kwargs.get("hop_length", sample_rate // 100)
        self.win_length = kwargs.get("win_length", sample_rate // 40)
        self.n_fft = kwargs.get("n_fft", sample_rate // 40)
        self.center = kwargs.get("center", True)

    def _cfg_trees(self):
        pre = {"sample_rate": self.sample_rate, "features": self.features, "hop_length": self.hop_length,
               "win_length": self.win_length, "n_fft": self.n_fft, "center": self.center}
        return pre, None, None

    def out_len(self, input_lengths: Tensor) -> Tensor:
        if self.center:
            return input_lengths.div(self.hop_length, rounding_mode="floor").add(1).long()
        return (input_lengths - self.win_length).div(self.hop_length, rounding_mode="floor").add(1).long()

    def forward(self, input_signal: Tensor, length: Tensor) -> Tuple[Tensor, Tensor]:
        return self.engine.frontend(input_signal, length)
