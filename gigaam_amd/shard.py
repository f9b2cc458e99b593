"""Multi-GPU sharding of the hot path: utterances are independent, so ranks own disjoint sets of
utterances / batches, weights are replicated, and the ONLY exchange is the final gather of the decoded
``(counts, ids, frames)``.  Host-side bookkeeping only -- which rank runs what, and how the gathered rows go back
to the caller's order; the exchange itself is ``gam_gather_ids`` (RCCL, include/gigaam_hip.h) on GPUs, or any
callable with the same contract (the world_size-2 gloo tests pass a torch.distributed one).

The reference has no multi-GPU inference path (its loop is gigaam/model.py:219-258, one device); the layout the
ranks assemble is the reference's collate layout (gigaam/utils.py:371-380) per batch.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
from torch import Tensor


def shard_range(n_items: int, rank: int, n_ranks: int) -> Tuple[int, int]:
    """Contiguous block of ceil(n/ranks) items per rank."""
    per = (n_items + n_ranks - 1) // n_ranks
    return min(n_items, rank * per), min(n_items, (rank + 1) * per)


def sorted_batches(lengths: Sequence[int], batch_size: int) -> List[List[int]]:
    """Indices sorted by length (longest first, ties by index) cut into consecutive batches: bounds the padding
    of every batch by the length spread inside it."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    return [order[i:i + batch_size] for i in range(0, len(order), batch_size)]


def deal(n_batches: int, rank: int, n_ranks: int, snake: bool = True) -> List[int]:
    """Batch indices of ``rank``.  ``snake``: boustrophedon order (0..R-1, R-1..0, ...) so that with batches sorted
    by cost every rank gets the same mix of long and short ones; otherwise plain round-robin."""
    out = []
    for j in range(n_batches):
        r = j % n_ranks
        if snake and (j // n_ranks) % 2 == 1:
            r = n_ranks - 1 - r
        if r == rank:
            out.append(j)
    return out


def lpt_deal(costs: Sequence[float], n_ranks: int) -> List[List[int]]:
    """Longest-processing-time-first dealing of ITEMS (not batches): items in order of falling cost (ties: lower index
    first), each to the rank with the smallest total so far (ties: lower rank).  Returns, per rank, its items in the order
    they were dealt, i.e. sorted by falling cost.  The classic 4/3-approximation of the balanced partition; with many
    items per rank the totals agree to within one small item."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * n_ranks
    out: List[List[int]] = [[] for _ in range(n_ranks)]
    for i in order:
        r = min(range(n_ranks), key=lambda q: (load[q], q))
        out[r].append(i)
        load[r] += float(costs[i])
    return out


def rank_batches(costs: Sequence[float], n_ranks: int, batch_size: int) -> List[List[List[int]]]:
    """Per rank, its ``lpt_deal`` share cut into consecutive batches of ``batch_size``: every rank gets the same amount of
    audio (not the same number of batches), and every batch holds chunks of neighbouring lengths (its padding is bounded
    by the length spread inside it).  Config 5 / longform sharding: the reference's loop (gigaam/model.py:219-258) batches
    in file order on one device; here the FILE ORDER is restored in the returned result only (unpack_results)."""
    return [[share[k:k + batch_size] for k in range(0, len(share), batch_size)] for share in lpt_deal(costs, n_ranks)]


def pack_results(rows: Sequence[Tuple[int, Sequence[int], Sequence[int]]], n_rows: int, cap: int):
    """Local decode results [(global_index, ids, frames)] -> fixed-shape buffers for the gather:
    index i32 [n_rows] (-1 = unused row), counts i32 [n_rows], ids / frames i32 [n_rows, cap]."""
    import numpy as np
    index = np.full((n_rows,), -1, dtype=np.int32)
    counts = np.zeros((n_rows,), dtype=np.int32)
    ids = np.zeros((n_rows, cap), dtype=np.int32)
    frames = np.zeros((n_rows, cap), dtype=np.int32)
    assert len(rows) <= n_rows
    for r, (g, i, f) in enumerate(rows):
        n = len(i)
        assert n <= cap and len(f) == n
        index[r], counts[r] = g, n
        ids[r, :n] = i
        frames[r, :n] = f
    return tuple(torch.from_numpy(x) for x in (index, counts, ids, frames))


def unpack_results(index: Tensor, counts: Tensor, ids: Tensor, frames: Tensor, n_total: int):
    """Gathered buffers (rank-major) -> [(ids, frames)] in global-index order; every index in [0, n_total) must
    appear exactly once."""
    index, counts, ids, frames = (t.cpu().numpy() for t in (index, counts, ids, frames))
    out: List[Optional[Tuple[List[int], List[int]]]] = [None] * n_total
    for r in range(index.shape[0]):
        g = int(index[r])
        if g < 0:
            continue
        if not 0 <= g < n_total or out[g] is not None:
            raise RuntimeError(f"utterance {g} gathered twice or out of range")
        n = int(counts[r])
        out[g] = (ids[r, :n].tolist(), frames[r, :n].tolist())
    missing = [g for g, o in enumerate(out) if o is None]
    if missing:
        raise RuntimeError(f"{len(missing)} utterances were never decoded (first: {missing[:5]})")
    return out


# --------------------------------------------------------------------------- the range flag across the exchange
# One rank's split-fp16 range flag (engine.Decoded.ext: the tail word of the buffer the decode's counts are a view of)
# must survive padding, the gather and the row selection.  It travels as ONE extra row of the exchanged buffers whose
# index word is FLAG_CLEAR or FLAG_SET (both negative: never mistaken for an utterance), so every rank learns every
# rank's flag in the exchange it performs anyway -- no extra host synchronisation, no attribute on a tensor view.
FLAG_CLEAR, FLAG_SET = -1, -2


def range_flag_of(dec) -> Optional[Tensor]:
    """The device word (i32 [1]) that received the range flag of a decode -- ``dec`` is the ``engine.Decoded`` object the decode
    call returned (an explicit field of it: nothing hangs off a tensor view any more) -- or None for anything else."""
    fw = getattr(dec, "flag_word", None)
    return fw() if callable(fw) else None


def append_flag_row(index: Tensor, counts: Tensor, ids: Tensor, frames: Tensor, flag: Optional[Tensor], rows: int):
    """Pad the local buffers to ``rows`` rows (index -1) and append the flag row: rows + 1 rows in all."""
    n, dev = ids.shape[0], ids.device
    assert n <= rows and index.shape[0] == rows
    pad = rows + 1 - n
    fi = torch.full((1,), FLAG_CLEAR, dtype=index.dtype, device=dev) if flag is None else (FLAG_CLEAR - (flag != 0).to(index.dtype)).reshape(1).to(dev)
    return (torch.cat([index, fi]), torch.cat([counts, counts.new_zeros((pad,))]),
            torch.cat([ids, ids.new_zeros((pad, ids.shape[1]))]), torch.cat([frames, frames.new_zeros((pad, frames.shape[1]))]))


def collect_gathered(gi: Tensor, gc: Tensor, gids: Tensor, gfr: Tensor):
    """Gathered buffers (rank-major, flag rows included) -> ([(ids, frames)] of the rows with index >= 0 in gathered
    order, their global indices, any_flag).  Two blocking copies: (index, counts), then the used part of ids / frames."""
    head = torch.stack([gi.to(torch.int32), gc.to(torch.int32)]).cpu()
    gi_h, gc_h = head[0].tolist(), head[1].tolist()
    flag = any(g == FLAG_SET for g in gi_h)
    sel = [r for r, g in enumerate(gi_h) if g >= 0]
    width = max((gc_h[r] for r in sel), default=0)
    if not sel:
        return [], [], flag
    st = torch.tensor(sel, dtype=torch.long, device=gids.device)
    both = torch.stack([gids.index_select(0, st)[:, :width], gfr.index_select(0, st)[:, :width]]).cpu()
    both_a = both.numpy()     # (numpy row slices -> lists: half the host time of per-row Tensor.tolist(), engine.collect)
    rows = [(both_a[0, k, :gc_h[r]].tolist(), both_a[1, k, :gc_h[r]].tolist()) for k, r in enumerate(sel)]
    return rows, [gi_h[r] for r in sel], flag


def run_sharded(batches: Sequence[Tuple[Tensor, Tensor, Sequence[int]]], decode_batch: Callable, rank: int, n_ranks: int,
                gather: Callable, cap: int, snake: bool = True, my_batches: Optional[Sequence[int]] = None,
                collect: Optional[Callable] = None, overlap_kw: bool = False):
    """Drive one rank's share of ``batches`` [(wav, len, global_indices)] through ``decode_batch(wav, len) ->
    [(ids, frames)]`` and gather everything: returns [(ids, frames)] for ALL utterances in global order (on every
    rank).  ``gather(index, counts, ids, frames) -> the same four, concatenated rank-major``; it is called exactly
    once, after the last local batch -- the path's one exchange.

    With ``collect``, ``decode_batch`` only LAUNCHES a batch (returns an opaque handle, no host sync) and
    ``collect(handle) -> [(ids, frames)]`` brings it to the host; batch n is launched before batch n-1 is collected,
    so the D2H wait and the host-side list building overlap the GPU's work on the next batch.  ``overlap_kw``: also call
    ``decode_batch(wav, len, overlap=<another batch of mine follows>)`` so that an RNN-T decode can run beside the next
    batch's encoder (model.launch_batch / engine.rnnt_greedy)."""
    mine = list(my_batches) if my_batches is not None else deal(len(batches), rank, n_ranks, snake)
    n_total = sum(len(b[2]) for b in batches)
    # every rank contributes the same number of rows (fixed-size all-gather): the largest share
    per_rank = max(sum(len(batches[j][2]) for j in deal(len(batches), r, n_ranks, snake)) for r in range(n_ranks))
    n_mine = sum(len(batches[j][2]) for j in mine)
    if n_mine > per_rank:   # (a caller-chosen my_batches that is not this rank's deal() share)
        raise ValueError(f"rank {rank}: my_batches holds {n_mine} utterances but the exchange is sized for {per_rank} per rank "
                         f"(deal(n_batches={len(batches)}, n_ranks={n_ranks}, snake={snake})); pass the deal() share or omit my_batches")
    rows = []

    def take(res, gidx):
        assert len(res) == len(gidx)
        rows.extend((int(g), i, f) for g, (i, f) in zip(gidx, res))

    pending = None
    for k, j in enumerate(mine):
        wav, wlen, gidx = batches[j]
        out = decode_batch(wav, wlen, overlap=k + 1 < len(mine)) if overlap_kw else decode_batch(wav, wlen)
        if collect is None:
            take(out, gidx)
        else:
            if pending is not None:
                take(collect(pending[0]), pending[1])
            pending = (out, gidx)
    if pending is not None:
        take(collect(pending[0]), pending[1])
    packed = pack_results(rows, per_rank, cap)
    return unpack_results(*gather(*packed), n_total)


# --------------------------------------------------------------------------- the exchange, through the C ABI
class HipComm:
    """``gam_comm`` (include/gigaam_hip.h): RCCL communicator behind the C ABI, one per rank.

    ``exchange_id(id_bytes_or_None) -> id_bytes`` is how the 128-byte RCCL id travels from rank 0 to the others;
    anything works (bench.py passes a torch.distributed broadcast, a reference-side binder could use a file)."""

    def __init__(self, rank: int, world: int, device: torch.device, exchange_id: Callable[[Optional[bytes]], bytes]):
        import ctypes as C

        from . import _lib
        self._C, self.lib = C, _lib.load_library()
        self.rank, self.world = rank, world
        self.device = torch.device(device)
        buf = C.create_string_buffer(128)
        err = None
        if rank == 0:
            rc = self.lib.gam_comm_unique_id(buf)
            if rc != 0:
                err = f"gam_comm_unique_id failed ({rc}): {self.lib.gam_comm_last_error(None).decode()}"
        # rank 0 always takes part in the exchange (an empty id tells the others it failed: nobody is left waiting)
        uid = exchange_id((b"" if err else buf.raw) if rank == 0 else None)
        if err or len(uid) != 128:
            raise _lib.GigaAMHipError(err or "rank 0 could not create an RCCL id")
        self._c = C.c_void_p()
        rc = self.lib.gam_comm_create(uid, rank, world, self.device.index or 0, C.byref(self._c))
        try:
            self._check(rc, "gam_comm_create")
        except Exception:
            self.close()        # a half-made communicator (the struct exists, ncclCommInitRank failed) is not leaked
            raise

    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            from ._lib import GigaAMHipError
            msg = self.lib.gam_comm_last_error(self._c)
            raise GigaAMHipError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")

    def gather(self, index: Optional[Tensor], counts: Tensor, ids: Tensor, frames: Tensor):
        """(index [rows] | None, counts [rows], ids [rows,cap], frames [rows,cap]) i32 on this rank's GPU ->
        the same, concatenated rank-major over all ranks; asynchronous on torch's current stream."""
        C = self._C
        dev = self.device
        cv = lambda t: None if t is None else t.to(device=dev, dtype=torch.int32).contiguous()  # noqa: E731
        index, counts, ids, frames = cv(index), cv(counts), cv(ids), cv(frames)
        rows, cap = ids.shape
        w = self.world
        a_index = None if index is None else torch.empty((w * rows,), dtype=torch.int32, device=dev)
        a_counts = torch.empty((w * rows,), dtype=torch.int32, device=dev)
        a_ids = torch.empty((w * rows, cap), dtype=torch.int32, device=dev)
        a_frames = torch.empty((w * rows, cap), dtype=torch.int32, device=dev)
        p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())  # noqa: E731
        with torch.cuda.device(dev):
            rc = self.lib.gam_gather_ids(self._c, p(index), p(counts), p(ids), p(frames), rows, cap, p(a_index), p(a_counts),
                                         p(a_ids), p(a_frames), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        self._check(rc, "gam_gather_ids")
        return a_index, a_counts, a_ids, a_frames

    def close(self) -> None:
        """ncclCommDestroy.  Explicit only: a communicator still alive at interpreter exit is left to the OS (destroying
        it from ``__del__`` during shutdown would run after torch has torn its HIP context down)."""
        if getattr(self, "_c", None):
            self.lib.gam_comm_destroy(self._c)
            self._c = None


def torch_gather(index: Optional[Tensor], counts: Tensor, ids: Tensor, frames: Tensor):
    """The same exchange over an initialised torch.distributed group (gloo in the CPU tests; kept as bench.py's
    ``--gather torch`` cross-check of the RCCL path).  World size 1: identity."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return index, counts, ids, frames
    outs = []
    for t in (index, counts, ids, frames):
        if t is None:
            outs.append(None)
            continue
        buf = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(buf, t.contiguous())
        outs.append(torch.cat(buf, dim=0))
    return tuple(outs)
