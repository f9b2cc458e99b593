"""CPU, world_size 2 over gloo: the N>1 harness path of bench.py -- shard bookkeeping,
the final gather of (counts, ids, frames) and the max-over-ranks timing reduction."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from common import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tp, b = 7, 3
    counts = torch.tensor([2 + rank, 0, 5], dtype=torch.int32)
    ids = torch.full((b, tp), -1, dtype=torch.int32)
    frames = torch.full((b, tp), -1, dtype=torch.int32)
    for i, c in enumerate(counts.tolist()):
        ids[i, :c] = torch.arange(c, dtype=torch.int32) + 10 * rank + i
        frames[i, :c] = torch.arange(c, dtype=torch.int32)
    g_counts, g_ids, g_frames = bench.gather_decoded(counts, ids, frames)
    tmax = bench.max_over_ranks(1.0 + rank)
    q.put((rank, g_counts.tolist(), g_ids.tolist(), g_frames.tolist(), tmax, bench.shard_range(10, rank, world)))
    dist.destroy_process_group()


def test_gather_and_timing_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, counts, ids, frames, tmax, shard in res:
        assert counts == [2, 0, 5, 3, 0, 5]            # rank-major order of the utterance shards
        assert ids[0][:2] == [0, 1] and ids[3][:3] == [10, 11, 12] and ids[5][:5] == [12, 13, 14, 15, 16]
        assert frames[2][:5] == [0, 1, 2, 3, 4]
        assert tmax == 2.0                              # MAX over ranks
    assert res[0][5] == (0, 5) and res[1][5] == (5, 10)
