"""GPU, world_size 2: the HIP hot path runs in BOTH ranks of a sharded job (BASELINE configs[4]) and the gathered
result equals the oracle. Rank 0 holds every codeblock's rate-matched LLRs (the single ingest point), `shard.scatter_units` deals
them block-cyclically (a block = the codeblocks of one transport block), each rank runs miphy_ldpc_rate_dematch_batch +
miphy_ldpc_decode_batch on its share, `shard.gather_results` brings iteration counts and hard bits back in unit order.

The GPU box has one card, so both ranks use cuda:0 and the process group is gloo (RCCL refuses two ranks on one device); the
sharding code is the one bench.py --gpus N runs over RCCL. The reference's analogue of "one instance per worker" is
lib/phy/upper/uplink_processor_concurrent.h:41-54."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BG, Z, NF, MOD, E, BLOCK = 1, 128, 8, 4, 3200, 3
K, N = 22 * Z, 66 * Z


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, llr_all, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import miphy
        from miphy import shard
        from miphy.ldpc import make_dec_descs
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        ctx = miphy.Context(0)
        n = llr_all.shape[0]
        payload = torch.from_numpy(llr_all).to(dev) if rank == 0 else None
        mine = shard.scatter_units(payload, n, 0, (E,), torch.int8, dev, block=BLOCK)
        m = mine.shape[0]
        rdm = np.zeros(m, dtype=miphy.LdpcRdmDesc)
        for i in range(m):
            rdm[i] = (BG, 0, MOD, 1, Z, NF, 0, E, i * E, i * N)
        sb = torch.full((m * N,), 77, dtype=torch.int8, device=dev)
        ctx.ldpc_rate_dematch_batch(rdm, mine.reshape(-1), sb)
        out = torch.zeros(m * (K // 8), dtype=torch.uint8, device=dev)
        it = torch.zeros(m, dtype=torch.int32, device=dev)
        ctx.ldpc_decode_batch(make_dec_descs(m, BG, Z, N, miphy.CRC24B, 6, NF), sb, out, it)
        torch.cuda.synchronize()
        rec = torch.cat([it.reshape(m, 1).to(torch.uint8), out.reshape(m, K // 8)], dim=1)  # per-unit result record
        allrec = shard.gather_results(rec, n, block=BLOCK)
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        q.put((rank, m, allrec.cpu().numpy(), float(t.item())))
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_hip_decode_in_two_ranks_matches_oracle():
    import torch
    import torch.multiprocessing as mp
    import oracle_lib as O
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    rng = np.random.default_rng(2024)
    n = 14  # not a multiple of world * block: ranks get 8 and 6 codeblocks
    llr_all = np.zeros((n, E), np.int8)
    exp = []
    for i in range(n):
        msg = rng.integers(0, 2, K, dtype=np.uint8)
        c = O.o_crc_bits(O.CRC24B, msg[:K - NF - 24])
        msg[K - NF - 24:K - NF] = [(c >> (23 - j)) & 1 for j in range(24)]
        msg[K - NF:] = 254
        cb = O.o_ldpc_encode(BG, Z, msg, N)
        rm = O.o_rate_match(0, MOD, 0, NF, cb, E)
        sigma = 0.45 if i % 5 else 1.6  # every fifth codeblock is undecodable: its record carries iteration count 0
        y = (1.0 - 2.0 * rm) + sigma * rng.standard_normal(E)
        llr_all[i] = np.round(np.clip(4 * y, -20, 20) / 20 * 120).astype(np.int8)
        sb = O.o_rate_dematch(0, MOD, 0, NF, 1, llr_all[i], np.full(N, 77, np.int8))
        exp.append(O.o_ldpc_decode(BG, Z, sb, NF, O.CRC24B, 6))
    world = 2
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, world, port, llr_all, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda x: x[0])
    [p.join(60) for p in procs]
    assert [r[1] for r in res] == [8, 6], "block-cyclic shares"
    assert all(r[3] == 2.0 for r in res), "MAX reduction of the elapsed time"
    for r in res:  # every rank holds the whole result in unit order
        rec = r[2]
        assert rec.shape == (n, 1 + K // 8)
        for i in range(n):
            it, bits = exp[i]
            assert rec[i, 0] == (it or 0), "codeblock %d: iterations %d vs oracle %s" % (i, rec[i, 0], it)
            assert np.array_equal(rec[i, 1:], bits), "codeblock %d: hard bits differ from the oracle" % i
    assert any(e[0] for e in exp) and not all(e[0] for e in exp)
