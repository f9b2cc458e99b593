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


class RNNTHead(_EngineModule):
    _prefix = "head."

    def __init__(self, decoder: Dict[str, int], joint: Dict[str, int]):
        super().__init__()
        self.decoder_cfg = dict(decoder)
        self.joint_cfg = dict(joint)

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
