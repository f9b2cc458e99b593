"""TEST INFRASTRUCTURE ONLY -- import shim for the *unmodified* reference modules.

Two places the reference can come from, in this order:

* ``/root/reference`` (the build container): the source tree itself;
* ``oracle/_ref`` (everywhere the snapshot travels, i.e. the GPU box): CPython bytecode of the same files, made by
  the committed recipe ``oracle/build_ref.py`` (``__graft_entry__.build()`` runs it) -- ``GIGAAM_REF_FORCE_BYTECODE=1``
  selects it in the build container too (tests/test_oracle_golden.py does, to prove both give the same numbers).

Who may use it: tests/ (golden generators, the live reference-vs-HIP sweeps), ``bench.py``'s ``cpu_baseline`` leg.
Nothing in the product path (``gigaam_amd/``) imports it.

The reference's modules import third-party packages that are absent from this image (torchaudio, hydra, omegaconf,
soundfile, onnxruntime, pyannote).  They are registered as stub modules before the import:

* ``hydra.utils.instantiate``: the ``_target_`` class-path constructor call (all the reference uses hydra for,
  ``gigaam/model.py:24-25,93-94``);
* ``omegaconf.DictConfig``: an attribute-access dict (``self.cfg.preprocessor``);
* ``torchaudio.transforms.MelSpectrogram``: a STAND-IN, the one piece here that is NOT reference code (row a1 stays
  parity-unpinned for exactly this reason): torchaudio's published ``Spectrogram`` -> ``MelScale`` contract restated on
  ``torch.stft`` (the native op torchaudio itself calls), with the buffers ``spectrogram.window`` / ``mel_scale.fb`` under
  torchaudio's own names so a checkpoint's state dict loads strictly.  With it the reference's ``FeatureExtractor``,
  ``SpecScaler``, ``GigaAM.forward`` and ``GigaAMASR._decode`` -- the body of ``transcribe()`` (``gigaam/model.py:126-140``)
  behind ``load_audio`` -- run unmodified.
"""
from __future__ import annotations

import importlib
import importlib.util
import json
import math
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("GIGAAM_REFERENCE_ROOT", "/root/reference")
BYTECODE_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def _bytecode_usable() -> bool:
    built = os.path.join(BYTECODE_ROOT, "BUILT.json")
    if not os.path.exists(built) or not os.path.exists(os.path.join(BYTECODE_ROOT, "gigaam", "encoder.pyc")):
        return False
    try:
        return json.load(open(built)).get("magic") == importlib.util.MAGIC_NUMBER.hex()
    except Exception:
        return False


def reference_kind():
    """'source' (/root/reference), 'bytecode' (oracle/_ref) or None."""
    force_bc = os.environ.get("GIGAAM_REF_FORCE_BYTECODE") == "1"
    if not force_bc and os.path.isdir(os.path.join(REFERENCE_ROOT, "gigaam")):
        return "source"
    if _bytecode_usable():
        return "bytecode"
    return None


def reference_available() -> bool:
    return reference_kind() is not None


def _stub(name: str, **attrs):
    if name in sys.modules:
        mod = sys.modules[name]
    else:
        mod = types.ModuleType(name)
        mod.__gam_stub__ = True
        sys.modules[name] = mod
    for k, v in attrs.items():
        if not hasattr(mod, k):
            setattr(mod, k, v)
    return mod


class AttrDict(dict):
    """Stand-in for omegaconf.DictConfig: keys readable as attributes, nested dicts wrapped on access."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v

    def get(self, k, default=None):  # noqa: D401
        v = dict.get(self, k, default)
        return AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v


def _instantiate(node, *args, **overrides):
    """hydra.utils.instantiate for the reference's use: ``cls = import(_target_); cls(**rest)``."""
    kw = dict(node)
    kw.update(overrides)
    target = kw.pop("_target_")
    mod, cls = target.rsplit(".", 1)
    plain = {k: (dict(v) if isinstance(v, dict) else v) for k, v in kw.items()}
    return getattr(importlib.import_module(mod), cls)(*args, **plain)


def _install_torchaudio_standin():
    """torchaudio.transforms.MelSpectrogram as published (defaults the reference does not override):
    Spectrogram(pad=0, window=periodic hann, power=2, normalized=False, center=<arg>, pad_mode='reflect', onesided) via
    torch.stft(...).abs().pow(2); MelScale(htk, norm=None, f_min=0, f_max=sr/2): mel = (spec^T @ fb)^T."""
    import torch
    from torch import nn

    class _Spectrogram(nn.Module):
        def __init__(self, n_fft, win_length, hop_length, center, pad_mode, power):
            super().__init__()
            self.n_fft, self.win_length, self.hop_length = n_fft, win_length, hop_length
            self.center, self.pad_mode, self.power = center, pad_mode, power
            self.register_buffer("window", torch.hann_window(win_length), persistent=True)

        def forward(self, waveform):
            shape = waveform.size()
            x = waveform.reshape(-1, shape[-1])
            spec = torch.stft(x, n_fft=self.n_fft, hop_length=self.hop_length, win_length=self.win_length, window=self.window,
                              center=self.center, pad_mode=self.pad_mode, normalized=False, onesided=True, return_complex=True)
            spec = spec.reshape(shape[:-1] + spec.shape[-2:])
            return spec.abs().pow(self.power)

    class _MelScale(nn.Module):
        def __init__(self, n_mels, sample_rate, f_min, f_max, n_stft):
            super().__init__()
            f_max = float(sample_rate // 2) if f_max is None else f_max
            hz2mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)  # noqa: E731
            all_freqs = torch.linspace(0, sample_rate // 2, n_stft)
            m_pts = torch.linspace(hz2mel(f_min), hz2mel(f_max), n_mels + 2)
            f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
            f_diff = f_pts[1:] - f_pts[:-1]
            slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
            down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
            up = slopes[:, 2:] / f_diff[1:]
            self.register_buffer("fb", torch.max(torch.zeros(1), torch.min(down, up)), persistent=True)

        def forward(self, specgram):
            return torch.matmul(specgram.transpose(-1, -2), self.fb).transpose(-1, -2)

    class MelSpectrogram(nn.Module):
        __gam_standin__ = True

        def __init__(self, sample_rate=16000, n_fft=400, win_length=None, hop_length=None, f_min=0.0, f_max=None,
                     pad=0, n_mels=128, power=2.0, normalized=False, center=True, pad_mode="reflect", **unused):
            super().__init__()
            win_length = n_fft if win_length is None else win_length
            hop_length = win_length // 2 if hop_length is None else hop_length
            assert pad == 0 and not normalized
            self.spectrogram = _Spectrogram(n_fft, win_length, hop_length, center, pad_mode, power)
            self.mel_scale = _MelScale(n_mels, sample_rate, f_min, f_max, n_fft // 2 + 1)

        def forward(self, waveform):
            return self.mel_scale(self.spectrogram(waveform))

    ta = _stub("torchaudio")
    tr = _stub("torchaudio.transforms")
    ta.transforms = tr
    ta.functional = _stub("torchaudio.functional")
    if getattr(ta, "__gam_stub__", False) and not hasattr(tr, "MelSpectrogram"):
        tr.MelSpectrogram = MelSpectrogram


_NS = None


def import_reference():
    """Return a namespace with the reference's hot-path modules (``encoder``, ``decoder``, ``decoding``, ``utils``,
    ``model``, ``preprocess``, ``timestamps_utils``, ``types``; ``kind`` says where they came from)."""
    global _NS
    if _NS is not None:
        return _NS
    kind = reference_kind()
    if kind is None:
        raise RuntimeError(f"reference not available: no tree at {REFERENCE_ROOT} and no usable bytecode under {BYTECODE_ROOT} "
                           "(python oracle/build_ref.py in the build container makes it)")
    _stub("soundfile")
    _install_torchaudio_standin()
    hy = _stub("hydra")
    hy.utils = _stub("hydra.utils", instantiate=_instantiate)
    _stub("omegaconf", DictConfig=AttrDict, ListConfig=list, OmegaConf=object)
    root = REFERENCE_ROOT if kind == "source" else BYTECODE_ROOT
    if root not in sys.path:
        sys.path.insert(0, root)
    ns = types.SimpleNamespace(kind=kind, root=root)
    for name in ("encoder", "decoder", "decoding", "utils", "preprocess", "types", "timestamps_utils", "model"):
        setattr(ns, name, importlib.import_module("gigaam." + name))
    origin = os.path.abspath(getattr(ns.encoder, "__file__", "") or "")
    assert origin.startswith(os.path.abspath(root)), f"gigaam.encoder came from {origin}, expected {root}"
    _NS = ns
    return ns


def import_onnx_twins():
    """The reference's torch-free statements of the two greedy decoders (gigaam/onnx_utils.py:39-54,73-161); its
    onnxruntime import is stubbed (the ORT sessions are never built here)."""
    import_reference()
    _stub("onnxruntime", InferenceSession=object, SessionOptions=object)
    return importlib.import_module("gigaam.onnx_utils")


def reference_model(ckpt):
    """The reference's own ``GigaAMASR`` / ``GigaAMEmo`` / ``GigaAM`` built from a ``{"cfg", "state_dict"}`` checkpoint the
    way ``gigaam.load_model`` does after the download (``gigaam/__init__.py:167-192``: class by head, ``load_state_dict``
    strict, ``.eval()``), on the CPU in fp32."""
    import torch  # noqa: F401
    ref = import_reference()
    cfg = AttrDict(ckpt["cfg"])
    head = ckpt["cfg"].get("head")
    if head is None:
        model = ref.model.GigaAM(cfg)
    elif head["_target_"].endswith("Linear"):
        model = ref.model.GigaAMEmo(cfg)
    else:
        model = ref.model.GigaAMASR(cfg)
    model.load_state_dict(ckpt["state_dict"], strict=True)
    return model.eval()
