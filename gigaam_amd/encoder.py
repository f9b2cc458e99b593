"""``cfg.encoder`` slot: ConformerEncoder on the HIP encoder (gam_encode).

Constructor signature and call contract follow reference
gigaam/encoder.py:510-526,605-647: ``(feat [B,feat_in,T], len [B]) ->
(encoded [B,d_model,T'], len i32 [B])``.
"""
from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor

from .preprocess import _EngineModule


class StridingSubsampling(_EngineModule):
    """``encoder.pre_encode`` view used by reference tests/test_batching.py:43-66:
    ``(x [B,T,feat], lengths) -> ([B,T',d_model], len i32)``; runs the stem only."""

    def __init__(self, owner: "ConformerEncoder"):
        super().__init__()
        object.__setattr__(self, "_owner", owner)

    def forward(self, x: Tensor, lengths: Tensor) -> Tuple[Tensor, Tensor]:
        _, elen, tok = self._owner.engine.encode(x.transpose(1, 2), lengths, n_layers_run=0, want_tokens=True)
        return tok, elen


class ConformerEncoder(_EngineModule):
    _prefix = "encoder."

    def __init__(
        self,
        feat_in: int = 64,
        n_layers: int = 16,
        d_model: int = 768,
        subsampling: str = "conv2d",
        subs_kernel_size: int = 3,
        subsampling_factor: int = 4,
        ff_expansion_factor: int = 4,
        self_attention_model: str = "rotary",
        n_heads: int = 16,
        pos_emb_max_len: int = 5000,
        conv_norm_type: str = "batch_norm",
        conv_kernel_size: int = 31,
        flash_attn: bool = False,
        activation_checkpointing: bool = False,
    ):
        super().__init__()
        assert self_attention_model in ["rotary", "rel_pos"], f"Not supported attn = {self_attention_model}"
        self.cfg = dict(
            feat_in=feat_in, n_layers=n_layers, d_model=d_model, subsampling=subsampling,
            subs_kernel_size=subs_kernel_size, subsampling_factor=subsampling_factor,
            ff_expansion_factor=ff_expansion_factor, self_attention_model=self_attention_model,
            n_heads=n_heads, pos_emb_max_len=pos_emb_max_len, conv_norm_type=conv_norm_type,
            conv_kernel_size=conv_kernel_size,
        )
        self.feat_in = feat_in
        # flash_attn / activation_checkpointing select torch code paths in the reference
        # (encoder.py:456-468,629-638); the HIP attention kernel is the only path here.
        self.flash_attn = flash_attn
        self.pre_encode = StridingSubsampling(self)

    def _cfg_trees(self):
        return None, self.cfg, None

    def forward_f32(self, audio_signal: Tensor, length: Tensor, host_lengths=None) -> Tuple[Tensor, Tensor]:
        """The kernels' own output: fp32 whatever the module's storage dtype is (what the heads of this package consume).
        ``host_lengths`` (not in the reference's signature): ``length`` as host integers -- a ragged batch then runs on its valid frames only
        (engine.encode); a ``length`` tensor that lives on the CPU is used for it automatically."""
        return self.engine.encode(audio_signal, length, host_lengths=host_lengths)

    def forward(self, audio_signal: Tensor, length: Tensor, host_lengths=None) -> Tuple[Tensor, Tensor]:
        enc, elen = self.forward_f32(audio_signal, length, host_lengths)
        if self._anchor.dtype != torch.float32:   # .half()-ed encoder (fp16_encoder=True): fp16 at the boundary
            enc = enc.to(self._anchor.dtype)
        return enc, elen

    def forward_layers(self, audio_signal: Tensor, length: Tensor, n_layers: int) -> Tuple[Tensor, Tensor]:
        """Test hook: token-major activations [B,T',d_model] after ``n_layers`` layers."""
        _, elen, tok = self.engine.encode(audio_signal, length, n_layers_run=n_layers, want_tokens=True)
        return tok, elen
