"""GPU: PDSCH modulator and PDSCH DM-RS mapping kernels through the C ABI: bit-exact against reference-produced
grids (tests/golden/pdsch_mod.npz) and against the oracle on random allocations; transmit -> receive loop with the demodulator."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _words(mask_bytes):
    w = np.zeros(5, dtype=np.uint64)
    for r in np.nonzero(mask_bytes)[0]:
        w[r >> 6] |= np.uint64(1) << np.uint64(r & 63)
    return w


def _mod_job(miphy, rnti, n_id, scaling, mod, port, start, nof, dm, type2, cdm, bwp_start, bwp_size, prb_list, reserved, nprb_grid, cw_off=0, grid_off=0):
    j = np.zeros(1, dtype=miphy.PdschModJob)[0]
    j["rnti"], j["n_id"], j["scaling"], j["mod"], j["port"], j["start_symbol"], j["nof_symbols"] = rnti, n_id, scaling, mod, port, start, nof
    j["dmrs_type"], j["nof_cdm_groups_without_data"], j["nof_reserved"] = 2 if type2 else 1, cdm, len(reserved)
    j["dmrs_symbols_mask"] = sum(1 << int(s) for s in np.nonzero(dm)[0])
    j["grid_nof_prb"], j["bwp_start_rb"], j["bwp_size_rb"] = nprb_grid, bwp_start, bwp_size
    rb = np.zeros(nprb_grid, np.uint8)
    rb[np.asarray(prb_list, dtype=int)] = 1
    j["rb_mask"] = _words(rb)
    for r, (pm, rm, sm) in enumerate(reserved):
        j["reserved"][r]["prb_mask"], j["reserved"][r]["re_mask"], j["reserved"][r]["symbols"] = _words(pm), rm, sm
    j["cw_offset"], j["grid_offset"] = cw_off, grid_off
    j["nof_bits"] = miphy.pdsch_mod_nof_re(j) * mod
    return j


def test_golden_grids(ctx):
    import torch
    import miphy
    d = np.load(os.path.join(GOLD, "pdsch_mod.npz"))
    for i in range(sum(1 for k in d.files if k.startswith("pm_cw_"))):
        rnti, n_id, scaling, mod, start, nof, type2, cdm, bwp_start, bwp_size, port, nprb_grid, ngp = d["pm_meta_%d" % i]
        ref, cw = d["pm_grid_%d" % i], d["pm_cw_%d" % i]
        res = [(d["pm_res_prb_%d" % i][r], int(d["pm_res_re_%d" % i][r, 0]), int(d["pm_res_re_%d" % i][r, 1])) for r in range(d["pm_res_prb_%d" % i].shape[0])]
        j = _mod_job(miphy, int(rnti), int(n_id), float(scaling), int(mod), int(port), int(start), int(nof), d["pm_dm_%d" % i], int(type2), int(cdm),
                     int(bwp_start), int(bwp_size), d["pm_prb_%d" % i], res, int(nprb_grid), cw_off=3)
        assert j["nof_bits"] == cw.size
        g = torch.zeros(ref.size, dtype=torch.complex64, device="cuda")
        ctx.pdsch_modulate_batch(np.array([j], dtype=miphy.PdschModJob), torch.from_numpy(np.concatenate([np.zeros(3, np.uint8), cw])).cuda(), g)
        torch.cuda.synchronize()
        assert np.array_equal(g.cpu().numpy().view(np.uint32), ref.reshape(-1).view(np.uint32)), i
    for i in range(sum(1 for k in d.files if k.startswith("dd_grid_"))):
        slot, ref_pt, type2, scr, nscid, amp, nports = d["dd_meta_%d" % i]
        ref, rb, sm = d["dd_grid_%d" % i], d["dd_rb_%d" % i], d["dd_sm_%d" % i]
        j = np.zeros(1, dtype=miphy.DmrsPdschJob)[0]
        j["slot_in_frame"], j["reference_point_k_rb"], j["scrambling_id"], j["amplitude"] = int(slot), int(ref_pt), int(scr), float(amp)
        j["dmrs_type"], j["n_scid"], j["nof_ports"] = 2 if type2 else 1, int(nscid), int(nports)
        j["ports"][:int(nports)] = np.arange(int(nports))
        j["symbols_mask"], j["grid_nof_prb"], j["rb_mask"] = sum(1 << int(s) for s in np.nonzero(sm)[0]), rb.size, _words(rb)
        g = torch.zeros(ref.size, dtype=torch.complex64, device="cuda")
        ctx.dmrs_pdsch_map_batch(np.array([j], dtype=miphy.DmrsPdschJob), g)
        torch.cuda.synchronize()
        assert np.array_equal(g.cpu().numpy().view(np.uint32), ref.reshape(-1).view(np.uint32)), i


@pytest.mark.parametrize("mod,nprb_grid,frac,start,nof,dsyms,type2,cdm,nres,scaling", [
    (8, 273, 1.0, 0, 14, (2,), 0, 2, 0, 1.0),
    (6, 106, 0.8, 1, 13, (2, 7, 11), 0, 1, 3, 0.5),
    (4, 52, 0.7, 2, 10, (3, 4), 1, 2, 4, 2.0),
    (2, 275, 0.9, 0, 14, (2,), 1, 3, 1, float("inf")),
    (1, 25, 0.6, 0, 14, (2, 11), 0, 2, 2, 1.0),
])
def test_random_allocations_match_oracle(ctx, mod, nprb_grid, frac, start, nof, dsyms, type2, cdm, nres, scaling):
    """The oracle's semantics (any ascending PRB set) on a batch of two transmissions sharing one grid buffer."""
    import torch
    import miphy
    rng = np.random.default_rng(17 * mod + nprb_grid)
    dm = np.zeros(14, np.uint8)
    dm[list(dsyms)] = 1
    jobs, cws, exp = [], [], []
    cw_off = 0
    for t in range(2):
        rb = (rng.uniform(size=nprb_grid) < frac).astype(np.uint8)
        rb[t] = 1
        pl = np.nonzero(rb)[0]
        reserved = [((rng.uniform(size=nprb_grid) < 0.4).astype(np.uint8), int(rng.integers(1, 4096)), int(rng.integers(1, 1 << 14))) for _ in range(nres)]
        rnti, n_id = int(rng.integers(1, 65536)), int(rng.integers(0, 1024))
        nre = O.pdsch_nof_re(pl, start, nof, dm, type2, cdm, 0, nprb_grid, reserved)
        cw = rng.integers(0, 2, nre * mod, dtype=np.uint8)
        g = np.zeros((2, 14, nprb_grid * 12), dtype=np.complex64)
        assert O.o_pdsch_modulate(rnti, n_id, scaling, 1, [mod], [cw], start, nof, dm, type2, cdm, 0, nprb_grid, pl, reserved, [t], nprb_grid, g) == nre
        j = _mod_job(miphy, rnti, n_id, scaling, mod, t, start, nof, dm, type2, cdm, 0, nprb_grid, pl, reserved, nprb_grid, cw_off=cw_off)
        assert j["nof_bits"] == cw.size
        jobs.append(j)
        cws.append(cw)
        exp.append(g)
        cw_off += cw.size
    gd = torch.zeros(2 * 14 * nprb_grid * 12, dtype=torch.complex64, device="cuda")
    ctx.pdsch_modulate_batch(np.array(jobs, dtype=miphy.PdschModJob), torch.from_numpy(np.concatenate(cws)).cuda(), gd)
    torch.cuda.synchronize()
    got = gd.cpu().numpy().reshape(2, 14, -1)
    want = exp[0] + exp[1]  # job t only writes port t
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_transmit_receive_loop(ctx):
    """Size-independent property: modulate + DM-RS -> (ideal channel) -> estimator + demodulator gives back the scrambled-and-
    descrambled codeword: the hard decisions of the LLRs are the transmitted bits."""
    import torch
    import miphy
    rng = np.random.default_rng(5)
    nprb, mod = 60, 6
    nsc = nprb * 12
    dm = np.zeros(14, np.uint8)
    dm[2] = 1
    rb = np.ones(nprb, np.uint8)
    j = _mod_job(miphy, 0x4601, 935, 1.0, mod, 0, 0, 14, dm, 0, 2, 0, nprb, np.arange(nprb), [], nprb)
    cw = rng.integers(0, 2, int(j["nof_bits"]), dtype=np.uint8)
    g = torch.zeros(14 * nsc, dtype=torch.complex64, device="cuda")
    ctx.pdsch_modulate_batch(np.array([j], dtype=miphy.PdschModJob), torch.from_numpy(cw).cuda(), g)
    dj = np.zeros(1, dtype=miphy.DmrsPdschJob)[0]
    dj["slot_in_frame"], dj["scrambling_id"], dj["amplitude"], dj["dmrs_type"], dj["nof_ports"] = 7, 42, 1.4125375, 1, 1
    dj["symbols_mask"], dj["grid_nof_prb"], dj["rb_mask"] = 1 << 2, nprb, _words(rb)
    ctx.dmrs_pdsch_map_batch(np.array([dj], dtype=miphy.DmrsPdschJob), g)
    cj = np.zeros(1, dtype=miphy.PuschChestJob)[0]
    cj["numerology"], cj["slot_in_frame"], cj["scrambling_id"], cj["scaling"] = 1, 7, 42, 1.4125375
    cj["nof_tx_layers"], cj["nof_rx_ports"], cj["first_symbol"], cj["nof_symbols"] = 1, 1, 0, 14
    cj["rx_ports"], cj["symbols_mask"], cj["grid_nof_prb"], cj["rb_mask"] = [0, 1, 2, 3], 1 << 2, nprb, _words(rb)
    ce = torch.zeros(14 * nsc, dtype=torch.complex64, device="cuda")
    sc = torch.zeros(5, dtype=torch.float32, device="cuda")
    g += torch.view_as_complex(torch.randn(14 * nsc, 2, device="cuda") * 0.01)
    ctx.dmrs_pusch_estimate_batch(np.array([cj], dtype=miphy.PuschChestJob), g, ce, sc)
    q = np.zeros(1, dtype=miphy.PuschDemodJob)[0]
    q["rnti"], q["n_id"], q["mod"], q["nof_rx_ports"], q["start_symbol"], q["nof_symbols"] = 0x4601, 935, mod, 1, 0, 14
    q["dmrs_type"], q["nof_cdm_groups_without_data"], q["ce_nof_symbols"], q["rx_ports"] = 1, 2, 14, [0, 1, 2, 3]
    q["dmrs_symbols_mask"], q["grid_nof_prb"], q["rb_mask"] = 1 << 2, nprb, _words(rb)
    q["nof_llr"] = miphy.pusch_demod_nof_llr(q)
    assert q["nof_llr"] == cw.size
    llr = torch.zeros(cw.size, dtype=torch.int8, device="cuda")
    ctx.pusch_demodulate_batch(np.array([q], dtype=miphy.PuschDemodJob), g, ce, sc, llr)
    torch.cuda.synchronize()
    hard = (llr.cpu().numpy() < 0).astype(np.uint8)
    assert np.array_equal(hard, cw)


def test_rejections(ctx):
    import torch
    import miphy
    dm = np.zeros(14, np.uint8)
    dm[2] = 1
    g = torch.zeros(14 * 120, dtype=torch.complex64, device="cuda")
    cw = torch.zeros(20000, dtype=torch.uint8, device="cuda")
    for bad in (dict(nof_bits=8), dict(mod=3), dict(dmrs_type=0), dict(nof_reserved=5), dict(nof_symbols=15)):
        j = _mod_job(miphy, 1, 2, 1.0, 4, 0, 0, 14, dm, 0, 2, 0, 10, np.arange(10), [], 10)
        for k, v in bad.items():
            j[k] = v
        with pytest.raises(RuntimeError):
            ctx.pdsch_modulate_batch(np.array([j], dtype=miphy.PdschModJob), cw, g)
