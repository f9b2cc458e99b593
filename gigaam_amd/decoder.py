"""``cfg.head`` slot: CTCHead / RNNTHead (reference gigaam/decoder.py:7-21,140-149)."""
from __future__ import annotations

from typing import Any, Dict

from torch import Tensor

from .preprocess import _EngineModule


class CTCHead(_EngineModule):
    _prefix = "head."

    def __init__(self, feat_in: int, num_classes: int):
        super().__init__()
        self.feat_in = feat_in
        self.num_classes = num_classes

    def _cfg_trees(self):
        return None, {"d_model": self.feat_in}, {"_target_": "CTCHead", "feat_in": self.feat_in, "num_classes": self.num_classes}

    def forward(self, encoder_output: Tensor) -> Tensor:
        """[B,feat_in,T'] -> log-probs [B,T',num_classes]."""
        return self.engine.ctc_head(encoder_output)


class _HeadPart:
    """``head.decoder`` / ``head.joint`` of the reference's RNNTHead (gigaam/decoder.py:24-149): the constructor arguments under
    the reference's attribute names, plus the per-step entry points the reference exposes -- ``RNNTDecoder.predict`` and
    ``RNNTJoint.joint`` / ``forward`` -- as calls into the library (``gam_rnnt_predict`` / ``gam_rnnt_joint``, r04).  The greedy
    decode of this path never goes through them (predictor, joint and the loop of gigaam/decoding.py:128-207 are ONE launch per
    batch, ``gam_rnnt_greedy``); they are for callers that take the head apart: a custom search, an export, a numerical check."""

    def __init__(self, kind: str, owner: "RNNTHead", **cfg: int):
        self._kind = kind
        self._owner = owner
        self.__dict__.update(cfg)

    def _wrong_part(self, name: str):
        raise AttributeError(f"RNNTHead.{self._kind} has no method {name!r} (it belongs to the other sub-module)")

    def predict(self, x, state, batch_size: int = 1):
        """RNNTDecoder.predict (gigaam/decoder.py:85-102): x i64 [B,U] labels or None, state (h, c) [L,B,H] or None ->
        (g [B,U,H], (h', c')).  U > 1 is evaluated step by step (the library call is one step)."""
        if self._kind != "decoder":
            self._wrong_part("predict")
        eng = self._owner.engine
        if x is None:
            g, state = eng.rnnt_predict(None, state, batch_size)
            return g.unsqueeze(1), state
        if x.dim() == 1:
            x = x.unsqueeze(1)
        outs = []
        for u in range(x.shape[1]):
            g, state = eng.rnnt_predict(x[:, u], state)
            outs.append(g)
        import torch
        return torch.stack(outs, dim=1), state

    def joint(self, encoder_out, decoder_out):
        """RNNTJoint.joint (gigaam/decoder.py:41-47): [B,T,enc_hidden], [B,U,pred_hidden] -> log-probs [B,T,U,V]."""
        if self._kind != "joint":
            self._wrong_part("joint")
        return self._owner.engine.rnnt_joint(encoder_out, decoder_out)

    def forward(self, *a):
        """RNNTJoint.forward(enc [B,enc_hidden,T], dec [B,pred_hidden,U]) (decoder.py:74-75) / RNNTDecoder.forward(x, h, c)
        (decoder.py:122-137: the export signature, state passed as two tensors)."""
        if self._kind == "joint":
            enc, dec = a
            return self.joint(enc.transpose(1, 2), dec.transpose(1, 2))
        x, h, c = a
        g, (h2, c2) = self.predict(x, (h, c))
        return g, h2, c2

    __call__ = forward


class RNNTHead(_EngineModule):
    _prefix = "head."

    def __init__(self, decoder: Dict[str, int], joint: Dict[str, int]):
        super().__init__()
        self.decoder_cfg = dict(decoder)
        self.joint_cfg = dict(joint)
        # (plain objects, not nn.Modules: kept out of the module tree)
        object.__setattr__(self, "decoder", _HeadPart("decoder", self, blank_id=self.decoder_cfg["num_classes"] - 1, **self.decoder_cfg))
        object.__setattr__(self, "joint", _HeadPart("joint", self, **self.joint_cfg))

    def _cfg_trees(self):
        head: Dict[str, Any] = {"_target_": "RNNTHead", "decoder": self.decoder_cfg, "joint": self.joint_cfg}
        return None, {"d_model": self.joint_cfg["enc_hidden"]}, head


class EmoHead(_EngineModule):
    """The emotion model's ``cfg.head`` (a ``torch.nn.Linear(in_features, out_features)`` in the
    reference's checkpoint; state_dict keys ``head.weight`` / ``head.bias``).  Called on the
    time-pooled encoder output by the reference (gigaam/model.py:277-282); here pooling, the
    projection and the softmax are one kernel behind ``gam_emo_probs``."""
    _prefix = "head."

    def __init__(self, in_features: int, out_features: int, bias: bool = True, **_: Any):
        super().__init__()
        assert bias, "the emotion head is a Linear with bias"
        self.in_features = in_features
        self.out_features = out_features

    def _cfg_trees(self):
        head = {"_target_": "torch.nn.Linear", "in_features": self.in_features, "out_features": self.out_features}
        return None, {"d_model": self.in_features}, head

    def probs(self, encoder_output: Tensor, lengths=None) -> Tensor:
        """[B,in_features,T'] -> softmax(Linear(mean_t encoder_output)) [B,out_features]."""
        return self.engine.emo_probs(encoder_output, lengths)
