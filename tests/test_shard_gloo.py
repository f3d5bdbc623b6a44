"""CPU, world_size 2 over gloo: the multi-GPU sharding helpers (block-cyclic unit assignment, ingest scatter, result
gather) are correct by construction -- every unit is owned exactly once, payloads arrive at their owner, results come back in
unit order. The data path itself has no collective (bench.py --gpus N only barriers and max-reduces the elapsed time)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from miphy import shard


def test_assignment_is_a_partition():
    for n in (1, 7, 38, 128, 1000):
        for world in (1, 2, 4, 8):
            for block in (1, 2, 38):
                seen = np.concatenate([shard.assign(n, world, r, block) for r in range(world)])
                assert sorted(seen.tolist()) == list(range(n))
                for u in range(n):
                    assert u in shard.assign(n, world, shard.owner(u, world, block), block)
    # weak scaling: every rank gets the same share when the unit count is a multiple of world*block
    assert all(len(shard.assign(8 * 38, 8, r, 38)) == 38 for r in range(8))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nof_units, block, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        payload = torch.arange(nof_units * 6, dtype=torch.int32).reshape(nof_units, 2, 3)
        # every rank makes the same call; only the source's payload is read
        mine = shard.scatter_units(payload if rank == 0 else None, nof_units, 0, (2, 3), torch.int32, torch.device("cpu"), block=block)
        idx = shard.assign(nof_units, world, rank, block)
        ok = torch.equal(mine, payload[torch.as_tensor(idx)])
        # a per-unit result record (here a checksum of the unit's payload; the HIP path in both ranks is covered by
        # tests/test_shard_hip_gpu.py) gathered back into unit order
        local = mine.reshape(len(idx), 6).sum(dim=1, keepdim=True).to(torch.int64)
        allres = shard.gather_results(local, nof_units, block=block)
        exp = payload.reshape(nof_units, 6).sum(dim=1, keepdim=True).to(torch.int64)
        ok = ok and torch.equal(allres, exp)
        # the timing reduction bench.py uses
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and float(t.item()) == float(world)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("nof_units,block", [(11, 1), (38, 4), (1, 1)])
def test_scatter_process_gather_world2(nof_units, block):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nof_units, block, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=60) for _ in range(world)]
    [p.join(30) for p in procs]
    assert sorted(res) == [(0, True), (1, True)]
