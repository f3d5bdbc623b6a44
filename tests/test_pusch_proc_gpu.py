"""GPU: fused PUSCH processor entry point (estimate + demodulate + decode in one call) on slots built by the device
transmit chain: the transport blocks come back, the HARQ retransmission path works, and the results equal those of the three
entry points called one by one (each of which has its own parity tests against the oracle)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RB_ALL = lambda nprb: [(0xFFFFFFFFFFFFFFFF if nprb >= 64 * (k + 1) else ((1 << max(0, nprb - 64 * k)) - 1)) for k in range(5)]


def _build_slot(ctx, miphy, torch, nprb, mod, tbs_bits, slot, rnti, n_id, scr, snr_db, rv, seed):
    rng = np.random.default_rng(seed)
    nsc, nre = nprb * 12, nprb * 156
    G = nre * mod
    tb = rng.integers(0, 256, tbs_bits // 8, dtype=np.uint8)
    bg = 1 if tbs_bits > 3824 else 2
    td = np.zeros(1, dtype=miphy.PdschTbDesc)
    td[0] = (bg, rv, mod, 1, 0, nre, tb.size, 0, 0)
    cw = torch.zeros(G, dtype=torch.uint8, device="cuda")
    ctx.pdsch_encode_batch(td, torch.from_numpy(tb).cuda(), cw)
    grid = torch.zeros(14 * nsc, dtype=torch.complex64, device="cuda")
    mj = np.zeros(1, dtype=miphy.PdschModJob)
    j = mj[0]
    j["rnti"], j["n_id"], j["scaling"], j["mod"], j["port"], j["start_symbol"], j["nof_symbols"] = rnti, n_id, 1.0, mod, 0, 0, 14
    j["dmrs_type"], j["nof_cdm_groups_without_data"], j["dmrs_symbols_mask"] = 1, 2, 1 << 2
    j["grid_nof_prb"], j["bwp_start_rb"], j["bwp_size_rb"], j["nof_bits"], j["rb_mask"] = nprb, 0, nprb, G, RB_ALL(nprb)
    ctx.pdsch_modulate_batch(mj, cw, grid)
    dj = np.zeros(1, dtype=miphy.DmrsPdschJob)
    q = dj[0]
    q["slot_in_frame"], q["scrambling_id"], q["amplitude"], q["dmrs_type"], q["nof_ports"] = slot, scr, 10 ** (3 / 20), 1, 1
    q["symbols_mask"], q["grid_nof_prb"], q["rb_mask"] = 1 << 2, nprb, RB_ALL(nprb)
    ctx.dmrs_pdsch_map_batch(dj, grid)
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    grid += torch.view_as_complex(torch.randn(14 * nsc, 2, device="cuda", generator=g) * (10 ** (-snr_db / 20) * 0.7071))
    return tb, bg, grid


def test_process_batch_recovers_transport_blocks_and_matches_the_separate_calls(ctx):
    import torch
    import miphy
    cases = [(273, 8, 319784, 33.0), (106, 6, 83976, 26.0), (52, 4, 20496, 21.0), (25, 2, 3848, 16.0)]  # SNRs with margin: the reference chain loses 12 % of the 16QAM blocks at 18 dB
    pdus = np.zeros(len(cases), dtype=miphy.PuschPdu)
    grids, tbs, goff, tboff, cboff = [], [], 0, 0, 0
    for i, (nprb, mod, tbs_bits, snr) in enumerate(cases):
        tb, bg, grid = _build_slot(ctx, miphy, torch, nprb, mod, tbs_bits, 7 + i, 0x4601 + i, 900 + i, 40 + i, snr, 0, 10 + i)
        p = pdus[i]
        p["numerology"], p["slot_in_frame"], p["rnti"], p["n_id"], p["dmrs_scrambling_id"] = 1, 7 + i, 0x4601 + i, 900 + i, 40 + i
        p["tb_bytes"], p["harq_cb_index"], p["mod"], p["nof_rx_ports"], p["start_symbol"], p["nof_symbols"] = tb.size, cboff, mod, 1, 0, 14
        p["bg"], p["rv"], p["new_data"], p["rx_ports"], p["use_early_stop"], p["nof_ldpc_iterations"] = bg, 0, 1, [0, 1, 2, 3], 1, 6
        p["dmrs_symbols_mask"], p["grid_nof_prb"], p["rb_mask"], p["grid_offset"], p["tb_offset"] = 1 << 2, nprb, RB_ALL(nprb), goff, tboff
        grids.append(grid)
        tbs.append(tb)
        goff += grid.numel()
        tboff += tb.size
        cboff += miphy.sch_segmentation(tb.size, bg).nof_cbs
    grid_d = torch.cat(grids)
    soft = torch.full((cboff * miphy.HARQ_CB_STRIDE,), 5, dtype=torch.int8, device="cuda")
    msgs = torch.zeros(cboff * miphy.HARQ_MSG_STRIDE, dtype=torch.uint8, device="cuda")
    crc = torch.zeros(cboff, dtype=torch.uint8, device="cuda")
    out = torch.zeros(tboff, dtype=torch.uint8, device="cuda")
    res = torch.zeros(len(cases) * miphy.PuschResult.itemsize, dtype=torch.uint8, device="cuda")
    sc = torch.zeros(len(cases) * 20, dtype=torch.float32, device="cuda")
    ctx.pusch_process_batch(pdus, grid_d, soft, msgs, crc, out, res, sc)
    torch.cuda.synchronize()
    r = res.cpu().numpy().view(miphy.PuschResult)
    o = out.cpu().numpy()
    scal = sc.cpu().numpy().reshape(len(cases), 4, 5)
    for i, tb in enumerate(tbs):
        assert r[i]["tb_crc_ok"] != 0, i
        t0 = int(pdus[i]["tb_offset"])
        assert np.array_equal(o[t0:t0 + tb.size], tb), i
        assert np.isfinite(scal[i, 0]).all() and scal[i, 0, 0] > 0 and scal[i, 0, 2] > 0  # RSRP and noise variance of port 0 are reported
    # the same through the three entry points
    for i, (nprb, mod, tbs_bits, snr) in enumerate(cases):
        nsc = nprb * 12
        cj = np.zeros(1, dtype=miphy.PuschChestJob)
        c = cj[0]
        c["numerology"], c["slot_in_frame"], c["scrambling_id"], c["scaling"] = 1, 7 + i, 40 + i, np.float32(10.0) ** np.float32(3.0 / 20.0)
        c["nof_tx_layers"], c["nof_rx_ports"], c["first_symbol"], c["nof_symbols"], c["rx_ports"] = 1, 1, 0, 14, [0, 1, 2, 3]
        c["symbols_mask"], c["grid_nof_prb"], c["rb_mask"] = 1 << 2, nprb, RB_ALL(nprb)
        ce = torch.zeros(14 * nsc, dtype=torch.complex64, device="cuda")
        s1 = torch.zeros(20, dtype=torch.float32, device="cuda")
        ctx.dmrs_pusch_estimate_batch(cj, grids[i], ce, s1)
        dq = np.zeros(1, dtype=miphy.PuschDemodJob)
        q = dq[0]
        q["rnti"], q["n_id"], q["mod"], q["nof_rx_ports"], q["start_symbol"], q["nof_symbols"] = 0x4601 + i, 900 + i, mod, 1, 0, 14
        q["dmrs_type"], q["nof_cdm_groups_without_data"], q["ce_nof_symbols"], q["rx_ports"] = 1, 2, 14, [0, 1, 2, 3]
        q["dmrs_symbols_mask"], q["grid_nof_prb"], q["rb_mask"] = 1 << 2, nprb, RB_ALL(nprb)
        q["nof_llr"] = miphy.pusch_demod_nof_llr(q)
        llr = torch.zeros(int(q["nof_llr"]), dtype=torch.int8, device="cuda")
        ctx.pusch_demodulate_batch(dq, grids[i], ce, s1, llr)
        td = np.zeros(1, dtype=miphy.PuschTbDesc)
        bg = int(pdus[i]["bg"])
        ncb = miphy.sch_segmentation(tbs[i].size, bg).nof_cbs
        td[0] = (bg, 0, mod, 1, 1, 1, 6, 0, nprb * 156, tbs[i].size, 0, 0, 0)
        so2 = torch.full((ncb * miphy.HARQ_CB_STRIDE,), 5, dtype=torch.int8, device="cuda")  # same stale content as the fused run
        ms2 = torch.zeros(ncb * miphy.HARQ_MSG_STRIDE, dtype=torch.uint8, device="cuda")
        cr2 = torch.zeros(ncb, dtype=torch.uint8, device="cuda")
        ou2 = torch.zeros(tbs[i].size, dtype=torch.uint8, device="cuda")
        re2 = torch.zeros(miphy.PuschResult.itemsize, dtype=torch.uint8, device="cuda")
        ctx.pusch_decode_batch(td, llr, so2, ms2, cr2, ou2, re2)
        torch.cuda.synchronize()
        r2 = re2.cpu().numpy().view(miphy.PuschResult)[0]
        assert (r2["tb_crc_ok"], r2["iters_min"], r2["iters_max"], r2["nof_decoded"]) == (r[i]["tb_crc_ok"], r[i]["iters_min"], r[i]["iters_max"],
                                                                                      r[i]["nof_decoded"]), i
        assert np.array_equal(s1.cpu().numpy()[:5], scal[i, 0]), i
        h0 = int(pdus[i]["harq_cb_index"])
        assert torch.equal(so2, soft[h0 * miphy.HARQ_CB_STRIDE:(h0 + ncb) * miphy.HARQ_CB_STRIDE]), i  # identical HARQ soft bits


def test_retransmission_through_the_processor(ctx):
    """HARQ through the fused entry point: at 21 dB the first 64QAM R=0.85 transmission fails; redundancy versions 2, 3, 1 are combined
    in the device-resident HARQ buffers until the transport block comes out. (With one DM-RS symbol the reference's estimator caps the
    SNR at 27 dB, so most LLRs saturate and combining needs more transmissions than an ideal receiver: the oracle decoder fed with
    the same LLRs behaves identically.)"""
    import torch
    import miphy
    nprb, mod, tbs_bits, SNR = 106, 6, 83976, 21.0
    soft = msgs = crc = out = None
    oks, tb = [], None
    for t, (slot, rv) in enumerate(((3, 0), (4, 2), (5, 3), (6, 1))):
        tb, bg, grid = _build_slot(ctx, miphy, torch, nprb, mod, tbs_bits, slot, 0x1234, 77, 9, SNR, rv, 99)
        if soft is None:
            ncb = miphy.sch_segmentation(tb.size, bg).nof_cbs
            soft = torch.zeros(ncb * miphy.HARQ_CB_STRIDE, dtype=torch.int8, device="cuda")
            msgs = torch.zeros(ncb * miphy.HARQ_MSG_STRIDE, dtype=torch.uint8, device="cuda")
            crc = torch.zeros(ncb, dtype=torch.uint8, device="cuda")
            out = torch.zeros(tb.size, dtype=torch.uint8, device="cuda")
        res = torch.zeros(miphy.PuschResult.itemsize, dtype=torch.uint8, device="cuda")
        sc = torch.zeros(20, dtype=torch.float32, device="cuda")
        pd = np.zeros(1, dtype=miphy.PuschPdu)
        p = pd[0]
        p["numerology"], p["slot_in_frame"], p["rnti"], p["n_id"], p["dmrs_scrambling_id"] = 1, slot, 0x1234, 77, 9
        p["tb_bytes"], p["mod"], p["nof_rx_ports"], p["start_symbol"], p["nof_symbols"] = tb.size, mod, 1, 0, 14
        p["bg"], p["rv"], p["new_data"], p["rx_ports"], p["use_early_stop"], p["nof_ldpc_iterations"] = bg, rv, 1 if t == 0 else 0, [0, 1, 2, 3], 1, 6
        p["dmrs_symbols_mask"], p["grid_nof_prb"], p["rb_mask"] = 1 << 2, nprb, RB_ALL(nprb)
        ctx.pusch_process_batch(pd, grid, soft, msgs, crc, out, res, sc)
        torch.cuda.synchronize()
        oks.append(int(res.cpu().numpy().view(miphy.PuschResult)[0]["tb_crc_ok"]))
        if oks[-1]:
            break
    assert oks[0] == 0 and oks[-1] == 1, oks
    assert np.array_equal(out.cpu().numpy(), tb)
