"""Result containers returned by ``transcribe*`` -- same fields and ``str()``
behaviour as the reference's gigaam/types.py:18-68 (boundary types)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterator, List, Optional


@dataclass
class Word:
    text: str
    start: float
    end: float


@dataclass
class TranscriptionResult:
    text: str
    words: Optional[List[Word]] = None

    def __str__(self) -> str:
        return self.text


@dataclass
class Segment:
    text: str
    start: float
    end: float
    words: Optional[List[Word]] = None


@dataclass
class LongformTranscriptionResult:
    segments: List[Segment]

    @property
    def words(self) -> List[Word]:
        return [w for seg in self.segments if seg.words for w in seg.words]

    @property
    def has_word_timestamps(self) -> bool:
        return len(self.segments) > 0 and self.segments[0].words is not None

    @property
    def text(self) -> str:
        return " ".join(seg.text for seg in self.segments)

    def __str__(self) -> str:
        return self.text

    def __iter__(self) -> Iterator[Segment]:
        return iter(self.segments)

    def __len__(self) -> int:
        return len(self.segments)
