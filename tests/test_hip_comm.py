"""GPU: the multi-GPU exchange behind the C ABI (gam_comm_* / gam_gather_ids, RCCL) on the one GPU a test box has:
a world of one rank must initialise RCCL, run the grouped all-gather on torch's current stream and hand back exactly
what went in.  (The N>1 bookkeeping is covered on CPU by tests/test_distributed_gloo.py.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_gather_ids_world_of_one():
    from gigaam_amd import shard
    dev = torch.device("cuda:0")
    comm = shard.HipComm(0, 1, dev, lambda uid: uid)
    g = torch.Generator().manual_seed(0)
    rows, cap = 5, 9
    index = torch.tensor([3, 0, -1, 7, 2], dtype=torch.int32)
    counts = torch.randint(0, cap, (rows,), generator=g, dtype=torch.int32)
    ids = torch.randint(0, 1000, (rows, cap), generator=g, dtype=torch.int32)
    frames = torch.randint(0, 500, (rows, cap), generator=g, dtype=torch.int32)
    for _ in range(3):
        gi, gc, gids, gfr = comm.gather(index, counts, ids, frames)
        torch.cuda.synchronize()
        assert torch.equal(gi.cpu(), index) and torch.equal(gc.cpu(), counts)
        assert torch.equal(gids.cpu(), ids) and torch.equal(gfr.cpu(), frames)
    gi, gc, gids, gfr = comm.gather(None, counts, ids, frames)
    assert gi is None and torch.equal(gids.cpu(), ids)
    # the whole sharded driver on one rank through the RCCL exchange
    batches = [(torch.zeros(2, 10), torch.tensor([10, 7]), [1, 0]), (torch.zeros(1, 5), torch.tensor([5]), [2])]
    res = shard.run_sharded(batches, lambda w, l: [([int(n)], [0]) for n in l.tolist()], 0, 1, comm.gather, cap=4)
    assert res == [([7], [0]), ([10], [0]), ([5], [0])]
    comm.close()
