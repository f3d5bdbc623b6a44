"""GPU: the device-resident HARQ pool under miphy_pusch_decode_batch. UEs reserve softbuffers by (rnti, harq), transmit,
fail, retransmit in later slots while other UEs come and go; every verdict, transport block and iteration count must equal
the oracle decoder's, which keeps its own per-TB softbuffer."""
import numpy as np
import pytest

from oracle_lib import OraclePuschDecoder, o_pdsch_encode, o_segmentation

pytestmark = pytest.mark.gpu


def noisy(cw, sigma, rng):
    y = (1.0 - 2.0 * (cw & 1)) + sigma * rng.standard_normal(cw.size)
    return np.round(np.clip(4 * y, -20, 20) / 20 * 120).astype(np.int8)


def test_pool_backed_harq(ctx):
    import torch
    import miphy
    rng = np.random.default_rng(77)
    pool = miphy.HarqPool(ctx, max_softbuffers=6, max_nof_codeblocks=40, expire_timeout_slots=16, numerology=1)
    soft_d, msgs_d, crc_d = pool.arrays()
    assert soft_d.shape == (6 * 52, miphy.HARQ_CB_STRIDE) and msgs_d.shape == (6 * 52, miphy.HARQ_MSG_STRIDE) and crc_d.shape == (6 * 52,)
    soft_d.fill_(33)  # stale garbage: a new-data transmission must not depend on it
    res_d = torch.zeros(8 * miphy.PuschResult.itemsize, dtype=torch.uint8, device="cuda")
    rvs = [0, 2, 3, 1]
    # (rnti, harq, bg, mod, nprb, tbs bits, sigma, first slot)
    ues = [(0x4601, 0, 1, 4, 106, 42016, 0.62, 0), (0x4601, 1, 2, 2, 106, 3848, 1.3, 1), (0x4602, 0, 1, 6, 106, 83976, 0.62, 2),
           (0x4603, 5, 2, 2, 273, 9984, 1.3, 3), (0x4604, 2, 1, 4, 106, 42016, 0.45, 9), (0x4605, 0, 2, 2, 4, 320, 1.0, 12)]
    state = []
    for rnti, harq, bg, mod, nprb, tbs_bits, sigma, s0 in ues:
        nsym = nprb * 156
        tb = rng.integers(0, 256, tbs_bits // 8, dtype=np.uint8)
        seg = o_segmentation(tbs_bits, bg, mod, 1, nsym)
        state.append(dict(rnti=rnti, harq=harq, bg=bg, mod=mod, nsym=nsym, tb=tb, ncb=seg.nof_cbs, sigma=sigma, next_slot=s0, tx=0, done=False,
                          od=OraclePuschDecoder(bg, mod, 0, 1, nsym, tbs_bits // 8)))
    retx_ok, first_cbs = 0, {}
    for slot in range(0, 48):
        pool.run_slot(slot)
        batch = []
        for u in state:
            if u["done"] or u["next_slot"] != slot:
                continue
            b, first = pool.reserve(slot, u["rnti"], u["harq"], u["ncb"])
            assert b >= 0, (slot, u["rnti"])
            pool.lock(b)
            first_cbs.setdefault((u["rnti"], u["harq"]), first)
            assert first_cbs[(u["rnti"], u["harq"])] == first  # the retransmission finds the same softbuffer
            batch.append((u, b, first))
        if not batch:
            continue
        d = np.zeros(len(batch), dtype=miphy.PuschTbDesc)
        llr_off, tb_off, chunks = 0, 0, []
        for i, (u, b, first) in enumerate(batch):
            rv = rvs[u["tx"]]
            u["llr"] = noisy(o_pdsch_encode(u["bg"], rv, u["mod"], 0, 1, u["nsym"], u["tb"]), u["sigma"], rng)
            d[i] = (u["bg"], rv, u["mod"], 1, 1 if u["tx"] == 0 else 0, 1, 6, 0, u["nsym"], u["tb"].size, first, llr_off, tb_off)
            chunks.append(u["llr"])
            llr_off += u["llr"].size
            tb_off += u["tb"].size
        tb_d = torch.full((tb_off,), 0xEE, dtype=torch.uint8, device="cuda")
        ctx.pusch_decode_batch(d, torch.from_numpy(np.concatenate(chunks)).cuda(), soft_d, msgs_d, crc_d, tb_d, res_d)
        torch.cuda.synchronize()
        res = res_d.cpu().numpy().view(miphy.PuschResult)
        tb_out = tb_d.cpu().numpy()
        for i, (u, b, first) in enumerate(batch):
            ok, tbo, mm = u["od"].decode(u["llr"], rvs[u["tx"]], u["tx"] == 0, 6, True)
            key = (slot, hex(u["rnti"]), u["tx"])
            assert bool(res[i]["tb_crc_ok"]) == ok, key
            assert (int(res[i]["iters_min"]), int(res[i]["iters_max"])) == mm, key
            # softbuffer contents: codeblock CRC flags as the oracle's softbuffer holds them
            assert np.array_equal(crc_d[first:first + u["ncb"]].cpu().numpy() != 0, np.asarray(u["od"].cb_crc, dtype=bool)), key
            o0 = int(d[i]["tb_offset"])
            if ok:
                assert np.array_equal(tb_out[o0:o0 + u["tb"].size], u["tb"]), key
                retx_ok += u["tx"] > 0
                pool.release(b)
                u["done"] = True
            else:
                pool.unlock(b)
                u["tx"] += 1
                u["next_slot"] = slot + 8
                assert u["tx"] < 4, key
    assert all(u["done"] for u in state) and retx_ok >= 3
    pool.run_slot(60)
    assert pool.free_codeblocks() == 40 and all(pool.info(i).state == miphy.HARQ_AVAILABLE for i in range(6))
    # the budget is the reference's: 40 codeblocks over all softbuffers
    assert pool.reserve(61, 1, 0, 30)[0] == 0 and pool.reserve(61, 2, 0, 11)[0] == -1 and pool.reserve(61, 2, 0, 10)[0] == 1
    pool.close()
