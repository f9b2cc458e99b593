"""Encoder frames -> word timestamps (reference gigaam/timestamps_utils.py:8-53).
Pure host-side string work over the exact ``frames`` the HIP decoders emit."""
from __future__ import annotations

from typing import List

from .decoding import Tokenizer
from .preprocess import SAMPLE_RATE
from .types import Word


def compute_frame_shift(audio_length_samples: int, seq_len: int) -> float:
    return audio_length_samples / SAMPLE_RATE / seq_len


def frames_to_words(tokenizer: Tokenizer, token_ids: List[int], token_frames: List[int], frame_shift: float) -> List[Word]:
    """A word ends at a space token (char vocab) or right before a piece that
    starts with the SentencePiece marker; start = first frame, end = last frame + 1."""
    words: List[Word] = []
    pieces: List[str] = []
    frames: List[int] = []

    def flush() -> None:
        text = "".join(pieces).strip()
        if text:
            words.append(Word(text=text, start=frames[0] * frame_shift, end=(frames[-1] + 1) * frame_shift))
        pieces.clear()
        frames.clear()

    for tok, frame in zip(token_ids, token_frames):
        piece = tokenizer.id_to_str(tok)
        if piece.startswith("▁"):
            flush()
            piece = piece[1:]
        elif piece == " ":
            flush()
            continue
        pieces.append(piece)
        frames.append(frame)
    flush()
    return words
