"""Pinned-memory, double-buffered batch feeder (SURVEY.md §8f NEXT-1).

The reference moves every longform batch to the device synchronously inside the loop
(gigaam/model.py:230-233: ``wav_pad.to(device)``).  Here the zero-padded batch layout of
``AudioDataset.collate`` (gigaam/utils.py:371-380) is assembled directly in a pinned host
buffer and copied on a side stream while the previous batch is being transcribed, so the
41 MB of a 32 x 20 s batch never sits on the critical path.
"""
from __future__ import annotations

from typing import Iterable, Iterator, List, Sequence, Tuple

import torch
from torch import Tensor


def collate_lengths(segments: Sequence[Tensor]) -> Tuple[int, Tensor]:
    lens = torch.tensor([int(s.shape[-1]) for s in segments], dtype=torch.int64)
    return int(lens.max()) if len(segments) else 0, lens


def collate(segments: Sequence[Tensor], out: Tensor = None) -> Tuple[Tensor, Tensor]:
    """``AudioDataset.collate`` layout (reference gigaam/utils.py:371-380): ``(batch f32 [B, max_len]
    zero-padded, lengths i64 [B])``.  ``out`` (optional) is a flat buffer with at least B*max_len elements --
    the feeder passes its pinned staging buffer so the batch is assembled in place."""
    lmax, lens = collate_lengths(segments)
    b = len(segments)
    if out is None:
        batch = torch.zeros((b, lmax), dtype=segments[0].dtype if b else torch.float32)
    else:
        batch = out[: b * lmax].view(b, lmax)
    # plain single-threaded memset / memcpy through numpy views: a torch op on a 20 MB CPU tensor fans out over every
    # OpenMP thread of the host, and on a box with a CPU quota those spinning threads get the whole process throttled
    # for tens of milliseconds -- seen as 50 ms holes in the middle of a batch's kernel launches (profiles/r02_config5_*)
    hb = batch.numpy()
    for j, c in enumerate(segments):
        src = c.reshape(-1)
        n = src.shape[0]
        hb[j, :n] = src.numpy() if src.device.type == "cpu" and not src.requires_grad else src.detach().cpu().numpy()
        if out is not None and n < lmax:
            hb[j, n:] = 0          # only the padding tail of a reused staging buffer is cleared (every byte is written once)
    return batch, lens


def batches(segments: Sequence[Tensor], batch_size: int) -> Iterator[List[Tensor]]:
    for i in range(0, len(segments), batch_size):
        yield list(segments[i:i + batch_size])


class BatchFeeder:
    """Iterate ``(wav_dev [B,L] f32 zero-padded, len_dev [B] i64)`` over ``segments`` in order.

    Two pinned host staging buffers AND two device buffers, all allocated once: batch n+1 is collated into pinned slot
    (n+1) % 2 and copied to device slot (n+1) % 2 on a side stream while the consumer's kernels of batch n run.  The
    yielded ``wav`` is a VIEW of a device slot, valid until the batch after next is requested: the slot is rewritten two
    batches later, after an event recorded on the consumer's stream when it asks for the next batch (i.e. after it has
    enqueued everything that reads the view; ``len`` is a fresh tensor and may be kept) -- no
    per-batch device allocation (a fresh ``torch.empty`` on a side stream every batch cost ~4 ms per 41 MB batch: the
    caching allocator cannot recycle a block whose last use is on another stream without waiting for it) and no host sync.

    LIFETIME CONTRACT of the yielded ``wav`` (also in INTEGRATION.md): (1) enqueue every kernel that reads it on the
    stream that is current when you ask for the NEXT batch -- that is where the "slot consumed" event is recorded; (2) do not
    keep it past the request for the batch after next: ``list(BatchFeeder(...))`` over three or more batches holds views whose
    slots have been overwritten, silently.  A consumer that wants to keep batches passes ``copy=True``: every yielded ``wav`` is
    then a private clone (made on the consumer's stream), at the cost of the per-batch allocation the slots exist to avoid."""

    def __init__(self, segments: Sequence[Tensor], batch_size: int, device: torch.device, copy: bool = False):
        self.copy = bool(copy)
        self.segments = segments
        self.batch_size = batch_size
        self.device = torch.device(device)
        from .engine import HipEngine
        self._copy_stream = HipEngine.aux_streams(self.device)[2]     # (a stream that cannot share a hardware queue with the consumer's)
        max_b = min(batch_size, max(1, len(segments)))
        max_l = max((int(s.shape[-1]) for s in segments), default=1)
        self._pin = [torch.empty((max_b * max_l,), dtype=torch.float32).pin_memory() for _ in range(2)]
        self._pin_len = [torch.empty((max_b,), dtype=torch.int64).pin_memory() for _ in range(2)]
        self._dev = [torch.empty((max_b * max_l,), dtype=torch.float32, device=self.device) for _ in range(2)]
        self.collate_seconds, self.batches_staged = 0.0, 0     # host time spent assembling batches (bench.py reports it)
        self.host_lengths: List[int] = []   # sample counts of the batch yielded last, as host integers (model.launch_batch(host_lengths=...):
                                            # a ragged batch then runs on its valid frames only, engine.encode)
        self._ready = [None, None]      # H2D completion event of the copy last issued from pinned slot i
        self._consumed = [None, None]   # recorded on the consumer's stream once it has enqueued the readers of device slot i

    def _stage(self, slot: int, chunk: List[Tensor]):
        b = len(chunk)
        if self._ready[slot] is not None:       # the copy issued two batches ago must have left this pinned buffer
            self._ready[slot].synchronize()
        import time
        t0 = time.perf_counter()
        host, lens = collate(chunk, out=self._pin[slot])      # contiguous pinned view of exactly this batch
        self.collate_seconds += time.perf_counter() - t0
        self.batches_staged += 1
        lmax = host.shape[1]
        self._pin_len[slot][:b] = lens
        with torch.cuda.stream(self._copy_stream):
            if self._consumed[slot] is not None:              # the kernels that read this device slot two batches ago
                self._copy_stream.wait_event(self._consumed[slot])
            wav = self._dev[slot][: b * lmax].view(b, lmax)   # contiguous device view of exactly this batch's shape
            wav.copy_(host, non_blocking=True)
            # the lengths stay a fresh (128-byte) tensor per batch: callers keep them in the handle they collect one batch later
            ln = self._pin_len[slot][:b].to(self.device, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self._copy_stream)
        self._ready[slot] = ready
        return wav, ln, ready, [int(v) for v in lens.tolist()]

    def __iter__(self) -> Iterator[Tuple[Tensor, Tensor]]:
        it = batches(self.segments, self.batch_size)
        slot = 0
        nxt = next(it, None)
        staged = self._stage(slot, nxt) if nxt is not None else None
        while staged is not None:
            wav, ln, ready, host_l = staged
            nxt = next(it, None)
            other = slot ^ 1
            if nxt is not None:
                staged = self._stage(other, nxt)
            else:
                staged = None
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ready)
            ln.record_stream(cur)
            self.host_lengths = host_l
            yield (wav.clone() if self.copy else wav), ln
            # the consumer is back: everything that reads this slot's views has been enqueued on its stream
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self.device))
            self._consumed[slot] = done
            slot = other
