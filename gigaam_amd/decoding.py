"""``cfg.decoding`` slot: greedy decoders (reference gigaam/decoding.py:10-207).

``decode(head, encoded [B,D,T'], lengths [B]) -> [(text, ids, frames)]``.  The
token work (argmax/collapse, the RNN-T predictor/joint loop) runs in one HIP
launch per batch; only the final ragged ids/frames cross PCIe, once.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import Tensor

from .decoder import CTCHead, RNNTHead


class Tokenizer:
    """Char-wise vocabulary or a SentencePiece model (reference decoding.py:10-44)."""

    def __init__(self, vocab: List[str], model_path: Optional[str] = None):
        self.charwise = model_path is None
        if self.charwise:
            self.vocab = vocab
        else:
            from sentencepiece import SentencePieceProcessor

            self.model = SentencePieceProcessor()
            self.model.load(model_path)

    def decode(self, tokens: List[int]) -> str:
        if self.charwise:
            return "".join(self.vocab[t] for t in tokens)
        return self.model.decode(tokens)

    def __len__(self) -> int:
        return len(self.vocab) if self.charwise else len(self.model)

    def id_to_str(self, token_id: int) -> str:
        return self.vocab[token_id] if self.charwise else self.model.IdToPiece(token_id)


class RangeOverflow(RuntimeError):
    """Raised by ``finish`` when the batch it collects set the split-fp16 range flag (include/gigaam_hip.h,
    gam_range_flag): an activation left fp16's range somewhere before this decode, so its ids are not to be trusted.
    The model shim catches it and repeats the work under GAM_GEMM_F32 (model.py); a caller that drives the decoders
    directly sees it as an error, never as silently wrong ids."""


def _ragged(dec, *rest) -> List[Tuple[List[int], List[int]]]:
    """What decode_device returned (an ``engine.Decoded``; three bare tensors are accepted) -> host lists (one D2H for
    counts + range flag, one for ids / frames)."""
    from .engine import HipEngine
    rows, flag = HipEngine.collect(dec, *rest[:2])
    if flag:
        raise RangeOverflow("activation beyond the split-fp16 GEMM range (repeat under GAM_GEMM_F32)")
    return rows


class CTCGreedyDecoding:
    def __init__(self, vocabulary: List[str], model_path: Optional[str] = None):
        self.tokenizer = Tokenizer(vocabulary, model_path)
        self.blank_id = len(self.tokenizer)

    @torch.inference_mode()
    def decode_device(self, head: CTCHead, encoded: Tensor, lengths: Tensor, overlap: bool = False):
        """The device half of ``decode``: (ids, frames, counts) i32 tensors on the GPU, no host sync -- a driver
        can launch the next batch before it looks at this one (``finish``).  ``overlap`` is accepted for signature symmetry
        with the RNN-T decoder and ignored: the CTC decode is one 8 us kernel."""
        c = head.num_classes
        assert c == len(self.tokenizer) + 1, f"Num classes {c} != len(vocab)+1 {len(self.tokenizer)+1}"
        return head.engine.ctc_greedy(encoded, lengths)

    def finish(self, dec, *rest) -> List[Tuple[str, List[int], List[int]]]:
        """``dec``: the object ``decode_device`` returned (pass it whole: it carries the range flag word and the decode's
        completion event as explicit fields)."""
        return [(self.tokenizer.decode(i), i, f) for i, f in _ragged(dec, *rest)]

    @torch.inference_mode()
    def decode(self, head: CTCHead, encoded: Tensor, lengths: Tensor) -> List[Tuple[str, List[int], List[int]]]:
        return self.finish(self.decode_device(head, encoded, lengths))


class RNNTGreedyDecoding:
    def __init__(self, vocabulary: List[str], model_path: Optional[str] = None, max_symbols_per_step: int = 10):
        self.tokenizer = Tokenizer(vocabulary, model_path)
        self.blank_id = len(self.tokenizer)
        self.max_symbols = max_symbols_per_step

    @torch.inference_mode()
    def decode_device(self, head: RNNTHead, encoded: Tensor, enc_len: Tensor, overlap: bool = False):
        """``overlap``: the caller is about to launch ANOTHER batch's frontend + encoder on the current stream -- run this
        latency-bound greedy loop beside it (decode side stream, small clusters: engine.HipEngine.rnnt_greedy) instead of in
        front of it.  Leave it False for a batch whose result is collected next (full-size clusters are faster alone)."""
        return head.engine.rnnt_greedy(encoded, enc_len, self.max_symbols, overlap=overlap)

    def finish(self, dec, *rest) -> List[Tuple[str, List[int], List[int]]]:
        """``dec``: the object ``decode_device`` returned (pass it whole: it carries the range flag word and the decode's
        completion event as explicit fields)."""
        return [(self.tokenizer.decode(i), i, f) for i, f in _ragged(dec, *rest)]

    @torch.inference_mode()
    def decode(self, head: RNNTHead, encoded: Tensor, enc_len: Tensor) -> List[Tuple[str, List[int], List[int]]]:
        return self.finish(self.decode_device(head, encoded, enc_len))
