"""gigaam_amd -- MI355X-native drop-in for GigaAM's inference hot path.

Public surface = the reference's (gigaam/__init__.py:15-22): ``load_model``,
``GigaAM``, ``GigaAMASR``, ``load_audio``, ``format_time``.  The log-mel frontend,
Conformer encoder and CTC / RNN-T greedy decoders run as hand-written gfx950 HIP
kernels in ``libgigaam_hip.so`` (C ABI: include/gigaam_hip.h).  There is no CPU
path: without the built library and a ROCm GPU every compute call raises.
"""
from __future__ import annotations

import hashlib
import os
from typing import Any, Optional, Union

import torch

from .model import GigaAM, GigaAMASR, GigaAMEmo
from .preprocess import load_audio

__all__ = ["GigaAM", "GigaAMASR", "GigaAMEmo", "load_audio", "format_time", "load_model", "model_from_checkpoint"]

_CACHE_DIR = os.path.expanduser("~/.cache/gigaam")
# md5 of the reference's published checkpoints (gigaam/__init__.py:28-41)
_MODEL_HASHES = {
    "emo": "7ce76f9535cb254488985057c0d33006",
    "v1_ctc": "f027f199e590a391d015aeede2e66174",
    "v1_rnnt": "02c758999bcdc6afcb2087ef256d47ef",
    "v1_ssl": "dc7f7b231f7f91c4968dc21910e7b396",
    "v2_ctc": "e00f59cb5d39624fb30d1786044795bf",
    "v2_rnnt": "547460139acfebd842323f59ed54ab54",
    "v2_ssl": "cd4cf819c8191a07b9d7edcad111668e",
    "v3_ctc": "73413e7be9c6a5935827bfab5c0dd678",
    "v3_rnnt": "0fd2c9a1ff66abd8d32a3a07f7592815",
    "v3_e2e_ctc": "367074d6498f426d960b25f49531cf68",
    "v3_e2e_rnnt": "2730de7545ac43ad256485a462b0a27a",
    "v3_ssl": "70cbf5ed7303a0ed242ddb257e9dc6a6",
}
_SHORT_NAMES = ["ctc", "rnnt", "e2e_ctc", "e2e_rnnt", "ssl"]


def format_time(seconds: float) -> str:
    """HH:MM:SS:cc (hours omitted when zero), reference gigaam/utils.py:68-80."""
    hours, rem = divmod(seconds, 3600)
    minutes, sec = divmod(rem, 60)
    whole = int(sec)
    cent = int((sec - whole) * 100)
    if int(hours) > 0:
        return f"{int(hours):02}:{int(minutes):02}:{whole:02}:{cent:02}"
    return f"{int(minutes):02}:{whole:02}:{cent:02}"


def _normalize_device(device: Optional[Union[str, torch.device]]) -> torch.device:
    if device is None:
        return torch.device("cuda" if torch.cuda.is_available() else "cpu")
    return torch.device(device) if isinstance(device, str) else device


def model_from_checkpoint(checkpoint: dict, device: Optional[Union[str, torch.device]] = None,
                          fp16_encoder: bool = False) -> Union[GigaAM, GigaAMASR, GigaAMEmo]:
    """Build a model from an in-memory ``{"cfg", "state_dict"}`` checkpoint (the layout
    of the reference's .ckpt files, gigaam/__init__.py:167-185)."""
    cfg = checkpoint["cfg"]
    name = cfg["model_name"] if isinstance(cfg, dict) else cfg.model_name
    if "ssl" in str(name):
        model = GigaAM(cfg)
    elif "emo" in str(name):   # gigaam/__init__.py:178-183
        model = GigaAMEmo(cfg)
    else:
        model = GigaAMASR(cfg)
    model.load_state_dict(checkpoint["state_dict"])
    model = model.eval()
    dev = _normalize_device(device)
    if fp16_encoder and dev.type != "cpu":   # gigaam/__init__.py:188-189
        model.encoder = model.encoder.half()
    return model.to(dev)


def load_model(model_name: str, fp16_encoder: bool = True, use_flash: Optional[bool] = False,
               device: Optional[Union[str, torch.device]] = None,
               download_root: Optional[str] = None) -> Union[GigaAM, GigaAMASR, GigaAMEmo]:
    """Same signature as the reference's ``load_model`` (gigaam/__init__.py:110-192).

    ``model_name`` is a model name (checkpoint expected at
    ``<download_root>/<name>.ckpt``; this build has no network access, so a missing
    file raises instead of downloading) or a path to a ``.ckpt`` holding
    ``{"cfg", "state_dict"}``.  ``fp16_encoder`` (default True, as in the reference) on a GPU
    gives the reference's storage contract -- ``_dtype`` float16, the waveform cast to float16 by
    ``prepare_wav``, ``encoded`` returned as float16 -- with fp32 arithmetic inside (the reference
    computes under fp16 autocast there); pass ``fp16_encoder=False`` for the fp32 contract of the
    reference's CPU path.  ``use_flash`` selects torch code paths in the reference; the HIP attention
    kernel is the only path here, so it is accepted and ignored."""
    del use_flash
    device_obj = _normalize_device(device)
    download_root = download_root or _CACHE_DIR
    local = os.path.expanduser(model_name)
    if os.path.isfile(local):
        ckpt = torch.load(local, map_location="cpu", weights_only=False)
        if "cfg" not in ckpt:  # fine-tuned Lightning checkpoint (gigaam/__init__.py:139-156)
            base = load_model(ckpt["hyper_parameters"]["model_name"], fp16_encoder=fp16_encoder, device=device_obj,
                              download_root=download_root)
            sd = {k: v for k, v in ckpt["state_dict"].items() if k.startswith(("preprocessor.", "encoder.", "head."))}
            base.load_state_dict(sd)
            return base
        return model_from_checkpoint(ckpt, device_obj, fp16_encoder)
    names = _SHORT_NAMES + list(_MODEL_HASHES.keys())
    if model_name not in names:
        raise ValueError(f"Model '{model_name}' not found. Available model names: {names}")
    if model_name in _SHORT_NAMES:
        model_name = f"v3_{model_name}"
    path = os.path.join(download_root, model_name + ".ckpt")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not found and this build cannot download checkpoints (no network); "
                                "place the reference's .ckpt there or pass a checkpoint path")
    digest = hashlib.md5(open(path, "rb").read()).hexdigest()
    assert digest == _MODEL_HASHES[model_name], f"Model checksum failed. Please run `rm {path}` and reload the model"
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    tok = os.path.join(download_root, model_name + "_tokenizer.model")
    if (model_name == "v1_rnnt" or "e2e" in model_name) and os.path.exists(tok):
        ckpt["cfg"].decoding.model_path = tok
    ckpt["cfg"].model_name = model_name
    return model_from_checkpoint(ckpt, device_obj, fp16_encoder)
