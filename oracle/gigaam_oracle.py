"""TEST INFRASTRUCTURE ONLY -- CPU fp32 oracle for the GigaAM hot path.

A functional restatement (torch CPU tensor ops, no nn.Module, no reference
imports) of SURVEY.md §8a rows a1-a14: log-mel frontend -> Conformer encoder ->
CTC / RNN-T heads and greedy decoders.  Every function cites the reference
file:line it follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; the product
(``gigaam_amd``) never does and fails loudly without its HIP library.

Pinning status
--------------
* a2-a14 (stem, encoder, heads, decoders): PINNED.  tests/golden/make_golden.py
  runs the reference's own unmodified modules (oracle/ref_shim.py) and this file
  on identical seeded weights/inputs in the build container, asserts agreement
  (<=2e-5 abs on encoder outputs, exact ids/frames), and commits the reference's
  outputs as fixtures under tests/golden/ which tests/test_oracle_golden.py
  re-checks on every run.  (r05: the reference itself travels too, as bytecode -- oracle/build_ref.py, oracle/ref_shim.py --
  and runs beside the kernels in tests/test_hip_vs_reference_live.py and bench.py's cpu_baseline leg.)
* a1 (FeatureExtractor = torchaudio.transforms.MelSpectrogram + log): PARITY
  UNPINNED.  torchaudio (pinned only as ``torchaudio>=2.6`` in the reference's
  pyproject.toml:30-33) is not installed and not under /root/reference, and
  the reference's tests that cover it need network + checkpoints.  The
  restatement follows torchaudio's published contract on
  ``torch.stft`` and is checked by analytic known-answer tests and against an
  independent fp64 scipy.fft implementation (tests/test_oracle_golden.py) --
  neither of which is the reference itself.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

SD = Dict[str, Tensor]


# =========================================================================== #
# a1  log-mel frontend  (reference gigaam/preprocess.py:43-98 + torchaudio)
# =========================================================================== #
def frontend_params(pre_cfg: dict) -> dict:
    """kwargs -> hop/win/n_fft/center exactly as preprocess.py:60-65."""
    sr = pre_cfg["sample_rate"]
    return {
        "sample_rate": sr,
        "n_mels": pre_cfg["features"],
        "hop_length": pre_cfg.get("hop_length", sr // 100),
        "win_length": pre_cfg.get("win_length", sr // 40),
        "n_fft": pre_cfg.get("n_fft", sr // 40),
        "center": pre_cfg.get("center", True),
    }


def feat_out_len(lengths: Tensor, fp: dict) -> Tensor:
    """preprocess.py:78-92."""
    if fp["center"]:
        return lengths.div(fp["hop_length"], rounding_mode="floor").add(1).long()
    return (lengths - fp["win_length"]).div(fp["hop_length"], rounding_mode="floor").add(1).long()


def log_mel(wav: Tensor, lengths: Tensor, pre_cfg: dict, window: Tensor, fb: Tensor) -> Tuple[Tensor, Tensor]:
    """wav f32 [B,L] -> (log-mel f32 [B,n_mels,T], len i64 [B]).

    torchaudio Spectrogram(power=2, center=<cfg>, pad_mode="reflect",
    onesided, normalized=False, window=periodic hann) -> MelScale(htk,
    norm=None): ``mel = (spec^T @ fb)^T`` -> SpecScaler
    ``log(clamp(x,1e-9,1e9))`` (preprocess.py:49-50,66-76,98)."""
    fp = frontend_params(pre_cfg)
    assert fp["win_length"] == fp["n_fft"] == window.numel()
    spec = torch.stft(
        wav.float(), n_fft=fp["n_fft"], hop_length=fp["hop_length"], win_length=fp["win_length"],
        window=window.float(), center=fp["center"], pad_mode="reflect", normalized=False,
        onesided=True, return_complex=True,
    )
    power = spec.real * spec.real + spec.imag * spec.imag  # |X|^2, [B, n_freq, T]
    mel = torch.matmul(power.transpose(-1, -2), fb.float()).transpose(-1, -2)
    return torch.log(mel.clamp(1e-9, 1e9)), feat_out_len(lengths, fp)


# =========================================================================== #
# a2  striding subsampling  (reference gigaam/encoder.py:32-130)
# =========================================================================== #
def calc_output_length(lengths: Tensor, kernel: int, stages: int) -> Tensor:
    """encoder.py:77-90 (float math, then int32)."""
    pad = (kernel - 1) // 2
    add_pad = 2 * pad - kernel
    x = lengths.to(torch.float)
    for _ in range(stages):
        x = torch.floor((x + add_pad) / 2 + 1.0)
    return x.to(torch.int32)


def _mask_time(x: Tensor, lengths: Tensor) -> Tensor:
    """encoder.py:92-109: zero frames t >= len along dim 2."""
    t = torch.arange(x.size(2))
    pad = t[None, :] >= lengths[:, None]
    pad = pad[:, None]
    if x.dim() == 4:
        pad = pad[..., None]
    return x.masked_fill(pad, 0.0)


def pre_encode(sd: SD, ecfg: dict, feat_btf: Tensor, lengths: Tensor, prefix: str = "encoder.") -> Tuple[Tensor, Tensor]:
    """[B,T,feat] -> ([B,T',d_model], len i32).  encoder.py:111-130."""
    k = ecfg["subs_kernel_size"]
    pad = (k - 1) // 2
    stages = int(math.log(ecfg["subsampling_factor"], 2))
    p = prefix + "pre_encode."
    conv2d = ecfg["subsampling"] == "conv2d"
    x = feat_btf.unsqueeze(1) if conv2d else feat_btf.transpose(1, 2)
    cur = lengths
    x = _mask_time(x, cur)
    for s in range(stages):
        w, b = sd[f"{p}conv.{2 * s}.weight"], sd[f"{p}conv.{2 * s}.bias"]
        x = F.conv2d(x, w, b, stride=2, padding=pad) if conv2d else F.conv1d(x, w, b, stride=2, padding=pad)
        cur = calc_output_length(cur, k, 1)
        x = _mask_time(x, cur)  # mask BEFORE the ReLU (encoder.py:119-123)
        x = F.relu(x)
    if conv2d:
        b_, _, t_, _ = x.shape
        x = F.linear(x.transpose(1, 2).reshape(b_, t_, -1), sd[p + "out.weight"], sd[p + "out.bias"])
    else:
        x = x.transpose(1, 2)
    return x, calc_output_length(lengths, k, stages)


# =========================================================================== #
# a3/a4  positional encodings  (encoder.py:307-361, utils.py:83-100)
# =========================================================================== #
def rotary_cos_sin(t: int, dim: int, base: int) -> Tuple[Tensor, Tensor]:
    """cos/sin [t,1,1,dim]; base = pos_emb_max_len (encoder.py:546-548)."""
    inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
    pos = torch.arange(t).type_as(inv_freq)
    freqs = torch.einsum("i,j->ij", pos, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos()[:, None, None, :], emb.sin()[:, None, None, :]


def _rotate_half(x: Tensor) -> Tensor:
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def rel_pos_emb(t: int, d_model: int, max_len: int) -> Tensor:
    """[1, 2t-1, d_model] slice of the sinusoid table (encoder.py:312-334)."""
    positions = torch.arange(max_len - 1, -max_len, -1).unsqueeze(1)
    pe = torch.zeros(positions.size(0), d_model)
    div_term = torch.exp(torch.arange(0, d_model, 2) * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(positions * div_term)
    pe[:, 1::2] = torch.cos(positions * div_term)
    pe = pe.unsqueeze(0)
    center = pe.size(1) // 2 + 1
    return pe[:, center - t: center + t - 1]


# =========================================================================== #
# a5-a8  Conformer layer  (encoder.py:133-498)
# =========================================================================== #
def _attention_core(q: Tensor, k: Tensor, v: Tensor, att_mask: Optional[Tensor], scores_bias: Optional[Tensor] = None) -> Tensor:
    """softmax(q k^T / sqrt(dk) [+bd]) v with masked keys excluded.
    q,k,v [B,H,T,dk]; att_mask [B,T,T] True = masked (encoder.py:620-624)."""
    dk = q.shape[-1]
    scores = torch.matmul(q, k.transpose(-2, -1))
    if scores_bias is not None:
        scores = scores + scores_bias
    scores = scores / math.sqrt(dk)
    if att_mask is not None:
        scores = scores.masked_fill(att_mask.unsqueeze(1), float("-inf"))
    # a fully masked (padded) query row has no defined value in the reference
    # either; keep it finite
    attn = torch.softmax(scores, dim=-1)
    attn = torch.nan_to_num(attn, nan=0.0)
    return torch.matmul(attn, v)


def self_attention(sd: SD, lp: str, ecfg: dict, x: Tensor, pos, att_mask: Optional[Tensor]) -> Tensor:
    b, t, d = x.shape
    h = ecfg["n_heads"]
    dk = d // h
    a = lp + "self_attn."
    lin = lambda name, inp: F.linear(inp, sd[a + f"linear_{name}.weight"], sd.get(a + f"linear_{name}.bias"))
    if ecfg["self_attention_model"] == "rotary":
        # RoPE on the (layer-normed) input BEFORE the projections, value path
        # un-rotated (encoder.py:244-256, utils.py:89-100)
        cos, sin = pos
        xh = x.transpose(0, 1).reshape(t, b, h, dk)
        xr = (xh * cos) + (_rotate_half(xh) * sin)
        xr = xr.reshape(t, b, d).transpose(0, 1)
        q = lin("q", xr).view(b, t, h, dk).transpose(1, 2)
        k = lin("k", xr).view(b, t, h, dk).transpose(1, 2)
        v = lin("v", x).view(b, t, h, dk).transpose(1, 2)
        ctx = _attention_core(q, k, v, att_mask)
    else:
        # rel_pos (encoder.py:208-228): bd[i,j] = (q_i + v_bias) . P(i-j)
        q = lin("q", x).view(b, t, h, dk)
        k = lin("k", x).view(b, t, h, dk).transpose(1, 2)
        v = lin("v", x).view(b, t, h, dk).transpose(1, 2)
        pvec = F.linear(pos, sd[a + "linear_pos.weight"]).view(1, -1, h, dk).transpose(1, 2)
        q_u = (q + sd[a + "pos_bias_u"]).transpose(1, 2)
        q_v = (q + sd[a + "pos_bias_v"]).transpose(1, 2)
        bd = torch.matmul(q_v, pvec.transpose(-2, -1))  # [B,H,T,2T-1]
        bb, hh, ql, pl = bd.shape
        bd = F.pad(bd, (1, 0)).view(bb, hh, -1, ql)[:, :, 1:].reshape(bb, hh, ql, pl)[..., :t]
        # reference masks with -10000 then zeroes (encoder.py:180-185); with
        # keys excluded outright the valid rows agree to fp32 round-off
        dk_ = q_u.shape[-1]
        scores = (torch.matmul(q_u, k.transpose(-2, -1)) + bd) / math.sqrt(dk_)
        if att_mask is not None:
            m = att_mask.unsqueeze(1)
            scores = scores.masked_fill(m, -10000.0)
            attn = torch.softmax(scores, dim=-1).masked_fill(m, 0.0)
        else:
            attn = torch.softmax(scores, dim=-1)
        ctx = torch.matmul(attn, v)
    ctx = ctx.transpose(1, 2).reshape(b, t, d)
    return lin("out", ctx)


def conv_module(sd: SD, lp: str, ecfg: dict, x: Tensor, pad_mask: Tensor) -> Tensor:
    """encoder.py:396-409.  x [B,T,D]; pad_mask [B,T] True = padded."""
    c = lp + "conv."
    d = x.shape[-1]
    k = ecfg["conv_kernel_size"]
    y = x.transpose(1, 2)
    y = F.conv1d(y, sd[c + "pointwise_conv1.weight"], sd[c + "pointwise_conv1.bias"])
    y = y[:, :d] * torch.sigmoid(y[:, d:])  # GLU over channels
    y = y.masked_fill(pad_mask.unsqueeze(1), 0.0)
    y = F.conv1d(y, sd[c + "depthwise_conv.weight"], sd[c + "depthwise_conv.bias"], padding=(k - 1) // 2, groups=d)
    if ecfg["conv_norm_type"] == "batch_norm":
        y = F.batch_norm(y, sd[c + "batch_norm.running_mean"], sd[c + "batch_norm.running_var"],
                         sd[c + "batch_norm.weight"], sd[c + "batch_norm.bias"], training=False, eps=1e-5)
    else:
        y = F.layer_norm(y.transpose(1, 2), (d,), sd[c + "batch_norm.weight"], sd[c + "batch_norm.bias"], 1e-5).transpose(1, 2)
    y = F.silu(y)
    y = F.conv1d(y, sd[c + "pointwise_conv2.weight"], sd[c + "pointwise_conv2.bias"])
    return y.transpose(1, 2)


def _ln(sd: SD, name: str, x: Tensor) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], 1e-5)


def _ffn(sd: SD, name: str, x: Tensor) -> Tensor:
    h = F.silu(F.linear(x, sd[name + ".linear1.weight"], sd[name + ".linear1.bias"]))
    return F.linear(h, sd[name + ".linear2.weight"], sd[name + ".linear2.bias"])


def conformer_layer(sd: SD, lp: str, ecfg: dict, x: Tensor, pos, att_mask, pad_mask) -> Tensor:
    """encoder.py:473-498 (macaron, fc_factor 0.5)."""
    r = x
    r = r + 0.5 * _ffn(sd, lp + "feed_forward1", _ln(sd, lp + "norm_feed_forward1", r))
    r = r + self_attention(sd, lp, ecfg, _ln(sd, lp + "norm_self_att", r), pos, att_mask)
    r = r + conv_module(sd, lp, ecfg, _ln(sd, lp + "norm_conv", r), pad_mask)
    r = r + 0.5 * _ffn(sd, lp + "feed_forward2", _ln(sd, lp + "norm_feed_forward2", r))
    return _ln(sd, lp + "norm_out", r)


# =========================================================================== #
# a9  encoder  (encoder.py:605-647)
# =========================================================================== #
def encoder_forward(sd: SD, ecfg: dict, feat: Tensor, lengths: Tensor, prefix: str = "encoder.",
                    stages: Optional[dict] = None) -> Tuple[Tensor, Tensor]:
    """feat [B,feat_in,T], len [B] -> (encoded [B,d_model,T'], len i32 [B]).
    ``stages`` (optional dict) receives per-stage tensors for kernel tests."""
    x, enc_len = pre_encode(sd, ecfg, feat.transpose(1, 2), lengths, prefix)
    if stages is not None:
        stages["pre_encode"] = x.clone()
    b, t, d = x.shape
    if ecfg["self_attention_model"] == "rotary":
        pos = rotary_cos_sin(t, d // ecfg["n_heads"], ecfg["pos_emb_max_len"])
    else:
        pos = rel_pos_emb(t, d, ecfg["pos_emb_max_len"])
    valid = torch.arange(t).expand(b, -1) < enc_len.unsqueeze(-1)
    att_mask = None
    if b > 1:  # encoder.py:620-624: no attention mask at batch 1
        att_mask = ~(valid.unsqueeze(1) & valid.unsqueeze(2))
    pad_mask = ~valid
    for i in range(ecfg["n_layers"]):
        x = conformer_layer(sd, f"{prefix}layers.{i}.", ecfg, x, pos, att_mask, pad_mask)
        if stages is not None:
            stages[f"layer{i}"] = x.clone()
    return x.transpose(1, 2), enc_len


# =========================================================================== #
# a10/a11  CTC head + greedy  (decoder.py:18-21, decoding.py:56-96)
# =========================================================================== #
def ctc_log_probs(sd: SD, encoded: Tensor) -> Tensor:
    w, b = sd["head.decoder_layers.0.weight"], sd["head.decoder_layers.0.bias"]
    return F.log_softmax(F.conv1d(encoded, w, b).transpose(1, 2), dim=-1)


def ctc_greedy(log_probs: Tensor, lengths: Tensor) -> List[Tuple[List[int], List[int]]]:
    """[B,T,V] -> per sample (ids, frames); blank = V-1 (decoding.py:54)."""
    labels = log_probs.argmax(dim=-1)
    b, t = labels.shape
    blank = log_probs.shape[-1] - 1
    lengths = lengths.clamp(min=0, max=t)
    out = []
    for i in range(b):
        ids: List[int] = []
        frames: List[int] = []
        prev = None
        for j in range(int(lengths[i])):
            lab = int(labels[i, j])
            if lab != blank and lab != prev:
                ids.append(lab)
                frames.append(j)
            prev = lab
        out.append((ids, frames))
    return out


# =========================================================================== #
# a12-a14  RNN-T head + greedy  (decoder.py:41-47,85-102, decoding.py:128-207)
# =========================================================================== #
def lstm_step(sd: SD, x: Tensor, h: Tensor, c: Tensor, layer: int = 0) -> Tuple[Tensor, Tensor]:
    """One nn.LSTM cell step, gate order i,f,g,o, two bias vectors."""
    p = "head.decoder.lstm."
    gates = (F.linear(x, sd[f"{p}weight_ih_l{layer}"], sd[f"{p}bias_ih_l{layer}"])
             + F.linear(h, sd[f"{p}weight_hh_l{layer}"], sd[f"{p}bias_hh_l{layer}"]))
    i, f, g, o = gates.chunk(4, dim=-1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return h2, c2


def rnnt_predict(sd: SD, label: Optional[int], state, n_layers: int = 1):
    """decoder.py:85-102 for one sample.  label None -> zero embedding and zero
    state.  Returns (g [ph], state=(h [layers,ph], c [layers,ph]))."""
    emb_w = sd["head.decoder.embed.weight"]
    ph = emb_w.shape[1]
    x = torch.zeros(ph) if label is None else emb_w[label]
    if state is None:
        state = (torch.zeros(n_layers, ph), torch.zeros(n_layers, ph))
    hs, cs = [], []
    for layer in range(n_layers):
        h2, c2 = lstm_step(sd, x, state[0][layer], state[1][layer], layer)
        hs.append(h2)
        cs.append(c2)
        x = h2
    return x, (torch.stack(hs), torch.stack(cs))


def rnnt_joint(sd: SD, f: Tensor, g: Tensor) -> Tensor:
    """decoder.py:41-47 for vectors f [enc_hidden], g [pred_hidden] -> log-probs [V]."""
    e = F.linear(f, sd["head.joint.enc.weight"], sd["head.joint.enc.bias"])
    p = F.linear(g, sd["head.joint.pred.weight"], sd["head.joint.pred.bias"])
    z = F.linear(F.relu(e + p), sd["head.joint.joint_net.1.weight"], sd["head.joint.joint_net.1.bias"])
    return F.log_softmax(z, dim=-1)


def rnnt_greedy(sd: SD, encoded: Tensor, enc_len: Tensor, max_symbols: int = 10, n_layers: int = 1,
                trace: Optional[list] = None) -> List[Tuple[List[int], List[int]]]:
    """Per-sample restatement of decoding.py:128-207 (the reference batches
    samples per step, but every sample's recurrence is independent).  State and
    last label are committed only on a non-blank emission; a sample with no
    state re-runs predict(None, None) every step.  ``trace`` collects
    (b, t, log_probs[V]) for every joint evaluation."""
    x = encoded.transpose(1, 2)
    b, t_max, _ = x.shape
    v = sd["head.joint.joint_net.1.weight"].shape[0]
    blank = v - 1
    out = []
    for i in range(b):
        ids: List[int] = []
        frames: List[int] = []
        label, state = None, None
        for t in range(min(int(enc_len[i]), t_max)):
            for _ in range(max_symbols):
                g, new_state = rnnt_predict(sd, label, state, n_layers)
                lp = rnnt_joint(sd, x[i, t], g)
                if trace is not None:
                    trace.append((i, t, lp.clone()))
                k = int(lp.argmax())
                if k == blank:
                    break
                ids.append(k)
                frames.append(t)
                label, state = k, new_state
        out.append((ids, frames))
    return out


# =========================================================================== #
# a15 glue: wav -> ids   (model.py:27-37,96-140)
# =========================================================================== #
def emo_probs(sd: SD, encoded: Tensor, enc_len: Optional[Tensor] = None) -> Tensor:
    """GigaAMEmo.get_probs after the encoder (reference gigaam/model.py:277-283; the export twin
    :291-293 writes the same thing as encoded.mean(-1)): avg_pool1d over the whole T' axis ->
    head (a Linear; keys head.weight / head.bias) -> softmax.  enc_len (not in the reference, which
    only ever sees one unpadded file) restricts the mean to each utterance's valid frames."""
    if enc_len is None:
        pooled = F.avg_pool1d(encoded, kernel_size=encoded.shape[-1]).squeeze(-1)
    else:
        m = (torch.arange(encoded.shape[-1])[None, :] < enc_len[:, None]).to(encoded.dtype)
        pooled = (encoded * m[:, None, :]).sum(-1) / enc_len.clamp(min=1)[:, None].to(encoded.dtype)
    return torch.softmax(F.linear(pooled, sd["head.weight"], sd["head.bias"]), dim=-1)


def transcribe_ids(ckpt: dict, wav: Tensor, lengths: Tensor):
    """Full CPU pipeline on a ``{"cfg","state_dict"}`` checkpoint.
    Returns (decoded [(ids, frames)], encoded, enc_len)."""
    cfg, sd = ckpt["cfg"], ckpt["state_dict"]
    feat, flen = log_mel(wav, lengths, cfg["preprocessor"],
                         sd["preprocessor.featurizer.0.spectrogram.window"],
                         sd["preprocessor.featurizer.0.mel_scale.fb"])
    encoded, enc_len = encoder_forward(sd, cfg["encoder"], feat, flen)
    head = cfg["head"]["_target_"]
    if head.endswith("CTCHead"):
        dec = ctc_greedy(ctc_log_probs(sd, encoded), enc_len)
    else:
        dec = rnnt_greedy(sd, encoded, enc_len, cfg["decoding"].get("max_symbols_per_step", 10),
                          cfg["head"]["decoder"]["pred_rnn_layers"])
    return dec, encoded, enc_len
