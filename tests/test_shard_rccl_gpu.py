"""GPU: the sharding helpers and the timing reduction of bench.py over RCCL (torch.distributed backend "nccl") with device
tensors. One rank -- the GPU box has one card -- so this checks that RCCL initialises on the MI355X and that the collectives the
multi-GPU path uses (broadcast, all_gather, all_reduce MAX, barrier) run on device memory; the multi-rank logic is covered on gloo
(tests/test_shard_gloo.py)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_collectives_on_device_single_rank(ctx):
    import torch
    import torch.distributed as dist
    from miphy import shard
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        dev = torch.device("cuda", 0)
        nof_units = 38
        payload = torch.arange(nof_units * 6, dtype=torch.int32, device=dev).reshape(nof_units, 2, 3)
        mine = shard.scatter_units(payload, nof_units, 0, (2, 3), torch.int32, dev, block=2)
        assert torch.equal(mine, payload)
        local = mine.reshape(nof_units, -1).sum(dim=1, keepdim=True).to(torch.int64)
        allres = shard.gather_results(local, nof_units, block=2)
        assert allres.is_cuda and torch.equal(allres, local)
        t = torch.tensor([1.25], dtype=torch.float64, device=dev)
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        assert float(t.item()) == 1.25
        assert np.array_equal(shard.assign(nof_units, 1, 0), np.arange(nof_units))
    finally:
        dist.destroy_process_group()
