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
