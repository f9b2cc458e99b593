"""Seeded synthetic GigaAM checkpoints (``{"cfg": ..., "state_dict": ...}``).

The real checkpoints live on a CDN (reference ``gigaam/__init__.py:24-41``) and
cannot be fetched here, so benchmarks and parity tests run on random-init
weights *of the exact architecture*: same ``cfg`` sub-trees (``preprocessor``,
``encoder``, ``head``, ``decoding`` with their hydra ``_target_`` class paths,
reference ``gigaam/model.py:24-25,93-94``) and the same ``state_dict`` key set
and shapes the reference modules produce.

Weights come from numpy's PCG64 (stable across platforms / torch versions), so
the GPU box regenerates bit-identical tensors from the seed alone and the
golden fixtures only need to store outputs.

Plain default-init weights make greedy decodes degenerate (CTC collapses to one
label, RNN-T emits ``max_symbols`` tokens on every frame).  The generator
therefore (a) uses a larger gain on the heads so argmax margins are far above
fp32 round-off and labels vary over time, and (b) biases the blank logit so the
blank ratio resembles a trained model.  ``describe_decode`` style checks live
in tests/golden/make_golden.py.
"""
from __future__ import annotations

import copy
import math
from typing import Any, Dict, List, Optional

import numpy as np
import torch

CHAR_VOCAB: List[str] = list(" абвгдежзийклмнопрстуфхцчшщъыьэюя")  # 33 symbols, blank = 33
assert len(CHAR_VOCAB) == 33


def _e2e_vocab(n: int) -> List[str]:
    """Stand-in for the SentencePiece model of the e2e checkpoints (not on disk)."""
    return [("▁" if i % 3 == 0 else "") + f"t{i}" for i in range(n)]


def _encoder_cfg(**over: Any) -> Dict[str, Any]:
    cfg = {
        "_target_": "gigaam.encoder.ConformerEncoder",
        "feat_in": 64,
        "n_layers": 16,
        "d_model": 768,
        "subsampling": "conv2d",
        "subs_kernel_size": 3,
        "subsampling_factor": 4,
        "ff_expansion_factor": 4,
        "self_attention_model": "rotary",
        "n_heads": 16,
        "pos_emb_max_len": 5000,
        "conv_norm_type": "batch_norm",
        "conv_kernel_size": 31,
        "flash_attn": False,
    }
    cfg.update(over)
    return cfg


def _v3_encoder_cfg(**over: Any) -> Dict[str, Any]:
    return _encoder_cfg(
        subsampling="conv1d",
        subs_kernel_size=5,
        conv_norm_type="layer_norm",
        conv_kernel_size=5,
        **over,
    )


def _pre_cfg(v3: bool) -> Dict[str, Any]:
    cfg: Dict[str, Any] = {
        "_target_": "gigaam.preprocess.FeatureExtractor",
        "sample_rate": 16000,
        "features": 64,
    }
    if v3:
        cfg.update(win_length=320, n_fft=320, hop_length=160, center=False)
    return cfg


EMO_NAMES = ["angry", "sad", "neutral", "positive"]


def _ctc_head(v: int) -> Dict[str, Any]:
    return {"_target_": "gigaam.decoder.CTCHead", "feat_in": 768, "num_classes": v}


def _rnnt_head(v: int) -> Dict[str, Any]:
    return {
        "_target_": "gigaam.decoder.RNNTHead",
        "decoder": {"pred_hidden": 320, "pred_rnn_layers": 1, "num_classes": v},
        "joint": {"enc_hidden": 768, "pred_hidden": 320, "joint_hidden": 320, "num_classes": v},
    }


def _ctc_decoding(vocab: List[str]) -> Dict[str, Any]:
    return {"_target_": "gigaam.decoding.CTCGreedyDecoding", "vocabulary": vocab}


def _rnnt_decoding(vocab: List[str]) -> Dict[str, Any]:
    return {
        "_target_": "gigaam.decoding.RNNTGreedyDecoding",
        "vocabulary": vocab,
        "max_symbols_per_step": 10,
    }


def model_cfg(model_name: str, **encoder_overrides: Any) -> Dict[str, Any]:
    """Config tree for one of the reference's model names (reference
    ``gigaam/__init__.py:28-41``).  Values follow SURVEY.md Appendix A."""
    name = model_name
    if name in ("ctc", "rnnt", "e2e_ctc", "e2e_rnnt", "ssl"):
        name = "v3_" + name  # reference gigaam/__init__.py:78-79
    v3 = name.startswith("v3")
    v1 = name.startswith("v1")
    if v3:
        enc = _v3_encoder_cfg(**encoder_overrides)
    elif v1:
        enc = _encoder_cfg(self_attention_model="rel_pos", **encoder_overrides)
    else:
        enc = _encoder_cfg(**encoder_overrides)
    cfg: Dict[str, Any] = {
        "model_name": name,
        "preprocessor": _pre_cfg(v3),
        "encoder": enc,
    }
    if name.endswith("ssl"):
        return cfg
    if name == "emo":   # emotion model: conv2d-stem encoder + Linear head (gigaam/model.py:262-285)
        cfg["head"] = {"_target_": "torch.nn.Linear", "in_features": 768, "out_features": len(EMO_NAMES)}
        cfg["id2name"] = {i: n for i, n in enumerate(EMO_NAMES)}
        return cfg
    e2e = "e2e" in name
    vocab = _e2e_vocab(256 if name.endswith("e2e_ctc") else 1024) if e2e else CHAR_VOCAB
    v = len(vocab) + 1
    if name.endswith("ctc"):
        cfg["head"] = _ctc_head(v)
        cfg["decoding"] = _ctc_decoding(vocab)
    elif name.endswith("rnnt"):
        cfg["head"] = _rnnt_head(v)
        cfg["decoding"] = _rnnt_decoding(vocab)
    else:
        raise ValueError(f"Model '{model_name}' not found.")
    return cfg


# --------------------------------------------------------------------------- #
# frontend buffers (what torchaudio's MelSpectrogram registers as buffers)
# --------------------------------------------------------------------------- #
def hann_window_periodic(n: int) -> np.ndarray:
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * math.pi * k / n)).astype(np.float32)


def mel_filterbank_htk(n_freqs: int, n_mels: int, sample_rate: int) -> np.ndarray:
    """[n_freqs, n_mels] triangular HTK filterbank, norm=None, f_min=0,
    f_max=sr/2 (torchaudio.functional.melscale_fbanks contract)."""
    f_max = sample_rate / 2.0
    all_freqs = np.linspace(0.0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + 0.0 / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = np.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    return fb.astype(np.float32)


# --------------------------------------------------------------------------- #
# state_dict
# --------------------------------------------------------------------------- #
BR = 0.5  # gain of every residual-branch output projection: keeps the identity path
# dominant (as in a trained net) so time structure survives 16 layers


class _Rng:
    def __init__(self, seed: int):
        self.g = np.random.Generator(np.random.PCG64(seed))

    def uniform(self, shape, bound: float) -> torch.Tensor:
        a = self.g.random(size=shape, dtype=np.float32)
        return torch.from_numpy((a * 2.0 - 1.0) * np.float32(bound))

    def normal(self, shape, std: float, mean: float = 0.0) -> torch.Tensor:
        a = self.g.standard_normal(size=shape, dtype=np.float32)
        return torch.from_numpy(a * np.float32(std) + np.float32(mean))


def _linear(sd, rng, prefix, out_f, in_f, shape=None, gain=1.0, bias=True, bias_gain=0.2):
    bound = gain / math.sqrt(in_f)
    sd[prefix + ".weight"] = rng.uniform(shape or (out_f, in_f), bound)
    if bias:
        sd[prefix + ".bias"] = rng.uniform((out_f,), bias_gain / math.sqrt(in_f))


def _norm(sd, rng, prefix, d):
    sd[prefix + ".weight"] = rng.normal((d,), 0.05, 1.0)
    sd[prefix + ".bias"] = rng.normal((d,), 0.02)


def make_state_dict(cfg: Dict[str, Any], seed: int = 0,
                    calib: Optional[torch.Tensor] = None, rnnt_blank_bias: Optional[float] = None) -> Dict[str, torch.Tensor]:
    rng = _Rng(seed)
    sd: Dict[str, torch.Tensor] = {}
    pre = cfg["preprocessor"]
    sr = pre["sample_rate"]
    n_fft = pre.get("n_fft", sr // 40)
    win = pre.get("win_length", sr // 40)
    sd["preprocessor.featurizer.0.spectrogram.window"] = torch.from_numpy(hann_window_periodic(win))
    sd["preprocessor.featurizer.0.mel_scale.fb"] = torch.from_numpy(
        mel_filterbank_htk(n_fft // 2 + 1, pre["features"], sr)
    )

    e = cfg["encoder"]
    d = e["d_model"]
    feat = e["feat_in"]
    ks = e["subs_kernel_size"]
    p = "encoder.pre_encode."
    if e["subsampling"] == "conv2d":
        _linear(sd, rng, p + "conv.0", d, ks * ks, shape=(d, 1, ks, ks), gain=1.5)
        _linear(sd, rng, p + "conv.2", d, d * ks * ks, shape=(d, d, ks, ks), gain=2.0)
        f_out = feat
        for _ in range(int(math.log2(e["subsampling_factor"]))):
            f_out = (f_out + 2 * ((ks - 1) // 2) - ks) // 2 + 1
        _linear(sd, rng, p + "out", d, d * f_out, gain=2.0)
    else:
        _linear(sd, rng, p + "conv.0", d, feat * ks, shape=(d, feat, ks), gain=1.5)
        _linear(sd, rng, p + "conv.2", d, d * ks, shape=(d, d, ks), gain=2.0)

    dff = d * e["ff_expansion_factor"]
    h = e["n_heads"]
    k = e["conv_kernel_size"]
    for i in range(e["n_layers"]):
        lp = f"encoder.layers.{i}."
        _norm(sd, rng, lp + "norm_feed_forward1", d)
        _linear(sd, rng, lp + "feed_forward1.linear1", dff, d, gain=1.5)
        _linear(sd, rng, lp + "feed_forward1.linear2", d, dff, gain=BR)
        _norm(sd, rng, lp + "norm_conv", d)
        _linear(sd, rng, lp + "conv.pointwise_conv1", 2 * d, d, shape=(2 * d, d, 1), gain=1.5)
        _linear(sd, rng, lp + "conv.depthwise_conv", d, k, shape=(d, 1, k), gain=1.5)
        if e["conv_norm_type"] == "batch_norm":
            _norm(sd, rng, lp + "conv.batch_norm", d)
            sd[lp + "conv.batch_norm.running_mean"] = rng.normal((d,), 0.05)
            sd[lp + "conv.batch_norm.running_var"] = rng.uniform((d,), 0.2) + 0.5
            sd[lp + "conv.batch_norm.num_batches_tracked"] = torch.tensor(1000, dtype=torch.long)
        else:
            _norm(sd, rng, lp + "conv.batch_norm", d)
        _linear(sd, rng, lp + "conv.pointwise_conv2", d, d, shape=(d, d, 1), gain=BR)
        _norm(sd, rng, lp + "norm_self_att", d)
        for nm in ("q", "k", "v", "out"):
            _linear(sd, rng, lp + f"self_attn.linear_{nm}", d, d, gain=2.0 if nm in "qk" else (BR if nm == "out" else 1.5))
        if e["self_attention_model"] == "rel_pos":
            _linear(sd, rng, lp + "self_attn.linear_pos", d, d, gain=2.0, bias=False)
            sd[lp + "self_attn.pos_bias_u"] = rng.normal((h, d // h), 0.1)
            sd[lp + "self_attn.pos_bias_v"] = rng.normal((h, d // h), 0.1)
        _norm(sd, rng, lp + "norm_feed_forward2", d)
        _linear(sd, rng, lp + "feed_forward2.linear1", dff, d, gain=1.5)
        _linear(sd, rng, lp + "feed_forward2.linear2", d, dff, gain=BR)
        _norm(sd, rng, lp + "norm_out", d)

    head = cfg.get("head")
    if head is None:
        return sd
    # ``calib`` ~ time-average of the encoder output (a fixed vector that
    # dominates a random-weight encoder).  Folding -W.calib into the head bias
    # makes the decode depend on the time-varying part, i.e. non-degenerate.
    c = torch.zeros(d) if calib is None else calib.to(torch.float32)
    if head["_target_"].endswith("Linear"):
        nc = head["out_features"]
        _linear(sd, rng, "head", nc, head["in_features"], gain=6.0)
        sd["head.bias"] -= sd["head.weight"] @ c
    elif head["_target_"].endswith("CTCHead"):
        v = head["num_classes"]
        # large gain: top-1/top-2 logit margins >> fp32 noise; mild blank bias
        _linear(sd, rng, "head.decoder_layers.0", v, head["feat_in"], shape=(v, head["feat_in"], 1), gain=6.0)
        sd["head.decoder_layers.0.bias"] -= sd["head.decoder_layers.0.weight"][:, :, 0] @ c
        sd["head.decoder_layers.0.bias"][v - 1] += 1.5
    else:
        dec, jn = head["decoder"], head["joint"]
        v, ph = dec["num_classes"], dec["pred_hidden"]
        emb = rng.normal((v, ph), 1.0)
        emb[v - 1] = 0.0  # padding_idx row (reference gigaam/decoder.py:82)
        sd["head.decoder.embed.weight"] = emb
        for layer in range(dec["pred_rnn_layers"]):
            bound = 1.5 / math.sqrt(ph)
            sd[f"head.decoder.lstm.weight_ih_l{layer}"] = rng.uniform((4 * ph, ph), bound)
            sd[f"head.decoder.lstm.weight_hh_l{layer}"] = rng.uniform((4 * ph, ph), bound)
            sd[f"head.decoder.lstm.bias_ih_l{layer}"] = rng.uniform((4 * ph,), bound)
            sd[f"head.decoder.lstm.bias_hh_l{layer}"] = rng.uniform((4 * ph,), bound)
        _linear(sd, rng, "head.joint.pred", jn["joint_hidden"], jn["pred_hidden"], gain=6.0)
        _linear(sd, rng, "head.joint.enc", jn["joint_hidden"], jn["enc_hidden"], gain=8.0)
        sd["head.joint.enc.bias"] -= sd["head.joint.enc.weight"] @ c
        _linear(sd, rng, "head.joint.joint_net.1", jn["num_classes"], jn["joint_hidden"], gain=4.0)
        # blank bias: a trained transducer emits blank on most frames
        # (default: emission-heavy, several symbols per frame -- exercises max_symbols_per_step;
        #  rnnt_blank_bias overrides it, e.g. to a speech-like rate of a few tokens per second)
        sd["head.joint.joint_net.1.bias"][v - 1] += (5.3 + 0.35 * math.log(v)) if rnnt_blank_bias is None else rnnt_blank_bias
    return sd


def calib_key(cfg: Dict[str, Any], seed: int) -> str:
    e = cfg["encoder"]
    return "{}_{}_{}{}_l{}_s{}".format(
        e["subsampling"], e["self_attention_model"], e["conv_norm_type"], e["conv_kernel_size"],
        e["n_layers"], seed)


def load_calib(cfg: Dict[str, Any], seed: int) -> Optional[torch.Tensor]:
    """Committed calibration vector (made by tests/golden/make_calib.py), or None."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth_calib", calib_key(cfg, seed) + ".npy")
    if os.path.exists(path):
        return torch.from_numpy(np.load(path))
    return None


def make_checkpoint(model_name: str, seed: int = 0, calib="auto", rnnt_blank_bias: Optional[float] = None,
                    pred_rnn_layers: Optional[int] = None, **encoder_overrides: Any) -> Dict[str, Any]:
    """``calib``: "auto" = committed vector if there is one, None = zeros, or a tensor.  ``pred_rnn_layers`` overrides the
    RNN-T predictor's LSTM depth (every published checkpoint uses 1)."""
    cfg = model_cfg(model_name, **encoder_overrides)
    if pred_rnn_layers is not None:
        cfg["head"]["decoder"]["pred_rnn_layers"] = int(pred_rnn_layers)
    if isinstance(calib, str):
        calib = load_calib(cfg, seed)
    return {"cfg": copy.deepcopy(cfg), "state_dict": make_state_dict(cfg, seed, calib, rnnt_blank_bias)}


def synth_audio(batch: int, seconds: float, seed: int = 0, sample_rate: int = 16000,
                lengths: Optional[List[int]] = None, index0: int = 0) -> tuple:
    """Seeded synthetic utterances with speech-like time structure: a random
    sequence of 60-260 ms "syllables" (harmonic stacks with per-syllable pitch,
    spectral tilt and amplitude, some noisy, some silent) plus a noise floor;
    harmonic tones + noise + taper in the spirit of reference
    tests/test_batching.py:15-25.  Returns (wav f32 [B,L], len i64 [B]);
    samples beyond ``lengths[b]`` are zero (AudioDataset.collate layout,
    reference gigaam/utils.py:371-380).  Row ``b`` is utterance ``index0 + b`` of the seed's stream, so a rank
    can generate its own slice of a global batch."""
    n = int(round(seconds * sample_rate))
    wav = np.zeros((batch, n), dtype=np.float32)
    lens = np.full((batch,), n, dtype=np.int64) if lengths is None else np.asarray(lengths, dtype=np.int64)
    for b in range(batch):
        rng = np.random.Generator(np.random.PCG64([seed, index0 + b]))
        lb = int(lens[b])
        sig = np.zeros(lb, dtype=np.float64)
        pos = 0
        while pos < lb:
            dur = int(rng.uniform(0.06, 0.26) * sample_rate)
            end = min(lb, pos + dur)
            m = end - pos
            kind = rng.random()
            if kind < 0.15:            # silence
                seg = np.zeros(m)
            else:
                t = np.arange(m, dtype=np.float64) / sample_rate
                f0 = 90.0 * 2.0 ** rng.uniform(0.0, 1.6)
                tilt = rng.uniform(0.3, 1.2)
                seg = np.zeros(m)
                for hrm in range(1, int(rng.integers(3, 12))):
                    if hrm * f0 > 7000:
                        break
                    seg += (hrm ** -tilt) * rng.uniform(0.3, 1.0) * np.sin(
                        2 * math.pi * hrm * f0 * t + rng.uniform(0, 2 * math.pi))
                if kind > 0.8:         # fricative-like noise burst
                    seg = 0.3 * seg + 0.5 * rng.standard_normal(m)
                seg *= rng.uniform(0.2, 1.0) * np.hanning(m + 2)[1:-1] ** 0.5
            sig[pos:end] = seg
            pos = end
        sig += 0.003 * rng.standard_normal(lb)
        peak = max(1e-6, float(np.abs(sig).max()))
        wav[b, :lb] = (0.5 * sig / peak).astype(np.float32)
    return torch.from_numpy(wav), torch.from_numpy(lens)
