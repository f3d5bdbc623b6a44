"""GPU: the fused PDSCH processor (miphy_pdsch_process_batch, pdsch_processor::process): transport blocks to grid REs for a
batch of PDUs, bit-exact against the oracle chain pdsch_encoder -> pdsch_modulator -> dmrs_pdsch_processor with the parameters
the reference processor derives (pdsch_processor_impl.cpp:198-305)."""
import ctypes

import numpy as np
import pytest

import oracle_lib as O
from test_pdsch_mod_gpu import _words

pytestmark = pytest.mark.gpu
_libm = ctypes.CDLL("libm.so.6")
_libm.powf.restype = ctypes.c_float
_libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]


def db_to_amplitude(x):  # convert_dB_to_amplitude (math_utils.h:101-104), single precision
    return float(_libm.powf(10.0, np.float32(x) / np.float32(20.0)))


# bg, mod, tbs bits, rv, grid PRBs, bwp (start, size), allocation (first, count), start symbol, nof symbols, DM-RS symbols, CDM groups,
# reserved patterns, port, grid ports, PRB0 reference, (dmrs dB, data dB), lbrm bytes
PDUS = [
    (1, 8, 319784, 0, 273, (0, 273), (0, 273), 0, 14, (2,), 2, 0, 0, 1, 0, (0.0, 0.0), 3168),
    (1, 6, 83976, 2, 106, (0, 106), (0, 106), 1, 13, (2, 7, 11), 1, 2, 1, 2, 1, (-3.0, 0.0), 3168),
    (2, 2, 3848, 3, 60, (5, 50), (10, 30), 2, 12, (2, 11), 2, 1, 0, 1, 1, (3.0, -1.5), 1200),
    (1, 4, 42016, 1, 120, (10, 100), (20, 80), 0, 14, (3,), 2, 3, 2, 3, 0, (0.0, 2.0), 3168),
    (2, 2, 320, 0, 25, (0, 25), (3, 4), 0, 14, (2,), 1, 0, 0, 1, 0, (0.0, 0.0), 400),
]


def test_batch_matches_oracle_chain(ctx):
    import torch
    import miphy
    rng = np.random.default_rng(321)
    pdus = np.zeros(len(PDUS), dtype=miphy.PdschPdu)
    tbs, want, tb_off, grid_off = [], [], 0, 0
    for i, (bg, mod, tbs_bits, rv, nprb, (bs, bz), (a0, an), start, nof, dsyms, cdm, nres, port, ngp, prb0, (ddb, xdb), lbrm) in enumerate(PDUS):
        dm = np.zeros(14, np.uint8)
        dm[list(dsyms)] = 1
        rb = np.zeros(nprb, np.uint8)
        rb[a0:a0 + an] = 1
        reserved = [((rng.uniform(size=nprb) < 0.3).astype(np.uint8), int(rng.integers(1, 4096)), int(rng.integers(1, 1 << 14))) for _ in range(nres)]
        tb = rng.integers(0, 256, tbs_bits // 8, dtype=np.uint8)
        rnti, n_id, scr, nscid, slot = int(rng.integers(1, 65536)), int(rng.integers(0, 1024)), int(rng.integers(0, 65536)), int(rng.integers(0, 2)), int(rng.integers(0, 20))
        p = pdus[i]
        p["slot_in_frame"], p["rnti"], p["n_id"], p["dmrs_scrambling_id"], p["tbs_lbrm_bytes"], p["tb_bytes"] = slot, rnti, n_id, scr, lbrm, tb.size
        p["ratio_pdsch_dmrs_to_sss_dB"], p["ratio_pdsch_data_to_sss_dB"] = ddb, xdb
        p["bg"], p["rv"], p["mod"], p["port"], p["start_symbol"], p["nof_symbols"] = bg, rv, mod, port, start, nof
        p["nof_cdm_groups_without_data"], p["n_scid"], p["ref_point_prb0"], p["nof_reserved"] = cdm, nscid, prb0, nres
        p["dmrs_symbols_mask"] = sum(1 << s for s in dsyms)
        p["grid_nof_prb"], p["bwp_start_rb"], p["bwp_size_rb"], p["rb_mask"] = nprb, bs, bz, _words(rb)
        for r, (pm, rm, sm) in enumerate(reserved):
            p["reserved"][r]["prb_mask"], p["reserved"][r]["re_mask"], p["reserved"][r]["symbols"] = _words(pm), rm, sm
        p["tb_offset"], p["grid_offset"] = tb_off, grid_off
        # the oracle chain with the parameters pdsch_processor_impl derives
        pl = np.nonzero(rb)[0]
        nre = O.pdsch_nof_re(pl, start, nof, dm, 0, cdm, bs, bz, reserved)
        assert miphy.pdsch_pdu_nof_re(p) == nre, i
        cw = O.o_pdsch_encode(bg, rv, mod, lbrm * 8, 1, nre, tb)
        g = np.zeros((ngp, 14, nprb * 12), dtype=np.complex64)
        assert O.o_pdsch_modulate(rnti, n_id, db_to_amplitude(-xdb), 1, [mod], [cw], start, nof, dm, 0, cdm, bs, bz, pl, reserved, [port], nprb, g) == nre
        O.o_dmrs_pdsch_map(slot, bs if prb0 else 0, 0, scr, nscid, db_to_amplitude(-ddb), dm, rb, [port], g)
        want.append(g)
        tbs.append(tb)
        tb_off += (tb.size + 15) // 16 * 16
        grid_off += g.size
    tb_all = np.zeros(tb_off, np.uint8)
    for p, tb in zip(pdus, tbs):
        tb_all[int(p["tb_offset"]):int(p["tb_offset"]) + tb.size] = tb
    gd = torch.zeros(grid_off, dtype=torch.complex64, device="cuda")
    ctx.pdsch_process_batch(pdus, torch.from_numpy(tb_all).cuda(), gd)
    torch.cuda.synchronize()
    got = gd.cpu().numpy()
    for i, (p, g) in enumerate(zip(pdus, want)):
        o = int(p["grid_offset"])
        assert np.array_equal(got[o:o + g.size].view(np.uint32), g.reshape(-1).view(np.uint32)), i
    # the prepared plan (descriptors uploaded once) gives the same grid, run after run
    plan = miphy.PdschProcessPlan(ctx, pdus)
    for _ in range(2):
        gp = torch.zeros(grid_off, dtype=torch.complex64, device="cuda")
        plan.run(torch.from_numpy(tb_all).cuda(), gp)
        torch.cuda.synchronize()
        assert np.array_equal(gp.cpu().numpy().view(np.uint32), got.view(np.uint32))
    plan.close()
    # a second call reuses the work buffer; one PDU alone gives the same REs
    gd2 = torch.zeros(want[2].size, dtype=torch.complex64, device="cuda")
    one = pdus[2:3].copy()
    one["grid_offset"], one["tb_offset"] = 0, 0
    ctx.pdsch_process_batch(one, torch.from_numpy(tbs[2]).cuda(), gd2)
    torch.cuda.synchronize()
    assert np.array_equal(gd2.cpu().numpy().view(np.uint32), want[2].reshape(-1).view(np.uint32))


def test_rejects_what_the_reference_asserts(ctx):
    import torch
    import miphy
    p = np.zeros(1, dtype=miphy.PdschPdu)
    p["tbs_lbrm_bytes"], p["tb_bytes"], p["bg"], p["mod"], p["nof_symbols"], p["nof_cdm_groups_without_data"] = 3168, 481, 2, 2, 14, 2
    p["dmrs_symbols_mask"], p["grid_nof_prb"], p["bwp_size_rb"] = 1 << 2, 52, 52
    p["rb_mask"][0][0] = (1 << 20) - 1
    tb = torch.zeros(512, dtype=torch.uint8, device="cuda")
    g = torch.zeros(14 * 52 * 12, dtype=torch.complex64, device="cuda")
    ctx.pdsch_process_batch(p, tb, g)  # valid as is
    for field, value, msg in [("dmrs_symbols_mask", 0, "DM-RS symbol mask"), ("start_symbol", 3, "outside the time allocation|exceeds the slot"),
                              ("tbs_lbrm_bytes", 0, "LBRM"), ("tbs_lbrm_bytes", 3169, "LBRM"), ("nof_cdm_groups_without_data", 3, "CDM groups"),
                              ("bg", 3, "base graph"), ("nof_reserved", 5, "reserved RE patterns")]:
        q = p.copy()
        q[field] = value
        with pytest.raises(RuntimeError, match=msg):
            ctx.pdsch_process_batch(q, tb, g)
    q = p.copy()
    q["rb_mask"][0][0] = 0
    with pytest.raises(RuntimeError, match="empty allocation"):
        ctx.pdsch_process_batch(q, tb, g)
    # the plan refuses the same PDUs when it is created
    with pytest.raises(RuntimeError, match="empty allocation"):
        miphy.PdschProcessPlan(ctx, q)
    q = p.copy()
    q["mod"] = 3
    with pytest.raises(RuntimeError, match="modulation"):
        miphy.PdschProcessPlan(ctx, q)
