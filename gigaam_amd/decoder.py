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
    """``head.decoder`` / ``head.joint`` of the reference's RNNTHead (gigaam/decoder.py:24-149) as attribute views: the
    constructor arguments under the reference's attribute names.  Their per-step entry points (``RNNTDecoder.predict``,
    ``RNNTJoint.joint``) do not exist on this path -- the predictor, the joint and the greedy loop that calls them
    (gigaam/decoding.py:128-207) are ONE launch per batch (``gam_rnnt_greedy``) -- and the only reference callers that
    take these sub-modules apart are the ONNX export (model.py:183-192) and training (train_utils/module.py:130-144),
    both outside this path: calling them says so instead of failing with an AttributeError."""

    def __init__(self, kind: str, **cfg: int):
        self._kind = kind
        self.__dict__.update(cfg)

    def _no_step(self, name: str):
        raise NotImplementedError(
            f"RNNTHead.{self._kind}.{name}: the MI355X path evaluates the predictor / joint inside gam_rnnt_greedy, one "
            "launch per batch (RNNTGreedyDecoding.decode); there is no per-step entry point to export or train through")

    def predict(self, *a, **k):
        self._no_step("predict")

    def joint(self, *a, **k):
        self._no_step("joint")

    def forward(self, *a, **k):
        self._no_step("forward")

    __call__ = forward


class RNNTHead(_EngineModule):
    _prefix = "head."

    def __init__(self, decoder: Dict[str, int], joint: Dict[str, int]):
        super().__init__()
        self.decoder_cfg = dict(decoder)
        self.joint_cfg = dict(joint)
        # (plain objects, not nn.Modules: kept out of the module tree)
        object.__setattr__(self, "decoder", _HeadPart("decoder", blank_id=self.decoder_cfg["num_classes"] - 1, **self.decoder_cfg))
        object.__setattr__(self, "joint", _HeadPart("joint", **self.joint_cfg))

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
