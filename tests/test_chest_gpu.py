"""GPU parity for the DM-RS PUSCH channel estimator vs the CPU oracle.

Tolerances (floating point; the reference's own vector test uses 5e-4 on every output, port_channel_estimator_test.cpp:114-169):
channel coefficients 1e-4 * max|h| (the reference interpolates by repeated float accumulation, the kernel in closed form),
RSRP / EPRE / noise / SNR 1e-4 relative, time alignment exact to one IDFT tap (1/(4096*scs))."""
import numpy as np
import pytest

from oracle_lib import o_dmrs_pusch_estimate, o_gold

pytestmark = pytest.mark.gpu


def make_case(rng, nprb_grid, alloc, nports, nl, dm_syms, numerology=1, slot=3, scr=77, nscid=0, scaling=1.0, delay=0.0, snr_db=25.0):
    """Builds a grid that really contains the DM-RS of a flat/2-tap channel with a delay (valid pilots), plus noise."""
    rb = np.zeros(nprb_grid, np.uint8)
    rb[alloc] = 1
    sm = np.zeros(14, np.uint8)
    sm[dm_syms] = 1
    nsc = nprb_grid * 12
    g = ((rng.standard_normal((nports, 14, nsc)) + 1j * rng.standard_normal((nports, 14, nsc))) * 0.05).astype(np.complex64)
    k = np.arange(nsc)
    for p in range(nports):
        h = (0.8 + 0.3j) * np.exp(-2j * np.pi * k * delay / 4096) * np.exp(1j * p) + 0.2 * np.exp(-2j * np.pi * k * (delay + 9) / 4096)
        for l in dm_syms:
            c_init = (((14 * slot + l + 1) * (2 * scr + 1)) % (1 << 31) * (1 << 17) + 2 * scr + nscid) % (1 << 31)
            c = o_gold(c_init, 0, 12 * nprb_grid)
            pil = ((1 - 2.0 * c[0::2]) + 1j * (1 - 2.0 * c[1::2])) / np.sqrt(2)  # one per (prb, q) counted from PRB 0
            for ly in range(nl):
                delta = (ly // 2) % 2
                w = np.ones(6 * nprb_grid)
                if ly % 2:
                    w[1::2] = -1  # applied on the *allocated* pilot index parity below
                idx = 0
                for r in range(nprb_grid):
                    if not rb[r]:
                        continue
                    for q in range(6):
                        wf = -1.0 if (ly % 2 and idx % 2) else 1.0
                        kk = r * 12 + 2 * q + delta
                        g[p, l, kk] += np.complex64(scaling * h[kk] * pil[r * 6 + q] * wf)
                        idx += 1
    return (numerology, slot, False, scr, nscid, scaling, sm, rb, 0, 14, nl, g)


def run(ctx, cases):
    import torch
    import miphy
    jobs = np.zeros(len(cases), dtype=miphy.PuschChestJob)
    grids, g_off, ce_off, sc_off = [], 0, 0, 0
    for i, (mu, slot, t2, scr, nscid, scaling, sm, rb, first, nof, nl, g) in enumerate(cases):
        nports, _, nsc = g.shape
        j = jobs[i]
        j["numerology"], j["slot_in_frame"], j["scrambling_id"], j["scaling"] = mu, slot, scr, scaling
        j["n_scid"], j["nof_tx_layers"], j["nof_rx_ports"], j["first_symbol"], j["nof_symbols"] = nscid, nl, nports, first, nof
        j["rx_ports"] = [0, 1, 2, 3]
        j["symbols_mask"] = sum(int(b) << l for l, b in enumerate(sm))
        j["grid_nof_prb"] = rb.size
        m = [0] * 5
        for r, b in enumerate(rb):
            if b:
                m[r >> 6] |= 1 << (r & 63)
        j["rb_mask"] = m
        j["grid_offset"], j["ce_offset"], j["scalars_offset"] = g_off, ce_off, sc_off
        grids.append(g.reshape(-1))
        g_off += g.size
        ce_off += nl * nports * (first + nof) * nsc
        sc_off += nports * nl * 5
    g_d = torch.from_numpy(np.concatenate(grids)).cuda()
    ce_d = torch.ones(ce_off, dtype=torch.complex64, device="cuda")
    sc_d = torch.zeros(sc_off, dtype=torch.float32, device="cuda")
    ctx.dmrs_pusch_estimate_batch(jobs, g_d, ce_d, sc_d)
    torch.cuda.synchronize()
    ce, sc = ce_d.cpu().numpy(), sc_d.cpu().numpy()
    oracle_cache = {}
    for i, a in enumerate(cases):
        mu, slot, t2, scr, nscid, scaling, sm, rb, first, nof, nl, g = a
        nports, _, nsc = g.shape
        if id(a) not in oracle_cache:  # (the large-batch test repeats the same case objects within one call)
            oracle_cache[id(a)] = o_dmrs_pusch_estimate(*a)
        exp_ce, exp_sc = oracle_cache[id(a)]
        got_ce = ce[int(jobs[i]["ce_offset"]):][:exp_ce.size].reshape(exp_ce.shape)
        got_sc = sc[int(jobs[i]["scalars_offset"]):][:exp_sc.size].reshape(exp_sc.shape)
        mask = np.repeat(rb.astype(bool), 12)
        err = np.abs(got_ce[..., mask] - exp_ce[..., mask]).max() / np.abs(exp_ce[..., mask]).max()
        assert err < 1e-4, (i, err)
        # unallocated PRBs are left untouched (they keep the caller's initial value)
        assert np.all(got_ce[..., ~mask] == 1.0)
        for k in range(4):
            rel = np.abs(got_sc[..., k] - exp_sc[..., k]) / (np.abs(exp_sc[..., k]) + 1e-30)
            assert rel.max() < 1e-4, (i, k, got_sc[..., k], exp_sc[..., k])
        tap = 1.0 / (4096 * 15000.0 * (1 << mu))
        assert np.abs(got_sc[..., 4] - exp_sc[..., 4]).max() <= 1.01 * tap, (i, got_sc[..., 4], exp_sc[..., 4])


def test_chest_configs(ctx):
    rng = np.random.default_rng(31)
    cases = [
        make_case(rng, 273, slice(0, 273), 1, 1, [2], delay=5.0),
        make_case(rng, 273, slice(0, 273), 2, 1, [2, 7, 11], delay=-7.0),
        make_case(rng, 106, slice(10, 60), 2, 2, [2, 11], scaling=0.7071, delay=20.0),
        make_case(rng, 52, [0, 1, 2, 10, 11, 30, 31, 32, 33], 1, 1, [3], slot=17, scr=1000, nscid=1, delay=3.0),
        make_case(rng, 25, slice(0, 25), 4, 4, [2, 3, 10, 11], numerology=0, slot=9, delay=-30.0),
        make_case(rng, 273, slice(100, 101), 1, 1, [2], delay=0.0),
    ]
    run(ctx, cases)


def test_chest_batch_that_fills_the_chip(ctx):
    """A launch that leaves most of the chip idle runs the time-alignment chain (IDFT + peak search) in a workgroup of its own next to the one that
    estimates and stores (what the small batches of the other tests get); a batch that fills the chip keeps both in one workgroup. Same cases, repeated
    until the launch takes the second form: identical results are required of both."""
    rng = np.random.default_rng(33)
    cases = [
        make_case(rng, 273, slice(0, 273), 1, 1, [2], delay=4.0),
        make_case(rng, 106, slice(10, 60), 2, 2, [2, 11], scaling=0.7071, delay=-12.0),
        make_case(rng, 52, [0, 1, 2, 10, 11, 30, 31, 32, 33], 1, 1, [3], slot=17, scr=1000, nscid=1, delay=3.0),
    ]
    run(ctx, cases)        # 3 jobs x 2 ports x 2 layers: the split form
    run(ctx, cases * 40)   # 120 jobs x 2 x 2 workgroups > half the CUs: one workgroup per (job, port, layer)


def test_chest_random_grid_like_benchmark(ctx):
    """pusch_processor_benchmark.cpp:536-555 fills the grid with N(0, 1/2) noise: no valid pilots, still deterministic."""
    rng = np.random.default_rng(32)
    cases = []
    for nports, nl, syms in ((1, 1, [2]), (2, 1, [2, 7, 11]), (4, 2, [2, 11])):
        rb = np.ones(273, np.uint8)
        sm = np.zeros(14, np.uint8)
        sm[syms] = 1
        g = ((rng.standard_normal((nports, 14, 273 * 12)) + 1j * rng.standard_normal((nports, 14, 273 * 12))) * np.sqrt(0.5)).astype(np.complex64)
        cases.append((1, 0, False, 0, 0, 1.0, sm, rb, 0, 14, nl, g))
    run(ctx, cases)


def _pilots(rb, l, slot, scr, nscid, nl):
    """DM-RS type-1 pilots of OFDM symbol l as the PUSCH estimator generates them (dmrs_helper.h:45-96): (nl, 6 * allocated PRBs) complex64."""
    c_init = (((14 * slot + l + 1) * (2 * scr + 1)) % (1 << 31) * (1 << 17) + 2 * scr + nscid) % (1 << 31)
    c = o_gold(c_init, 0, 12 * rb.size).astype(np.float32)
    amp = np.float32(0.70710678118654752440)
    full = (amp * (np.float32(1) - np.float32(2) * c[0::2])) + 1j * (amp * (np.float32(1) - np.float32(2) * c[1::2]))
    sel = np.concatenate([np.arange(r * 6, r * 6 + 6) for r in np.nonzero(rb)[0]])
    p = full[sel].astype(np.complex64)
    out = np.zeros((nl, p.size), np.complex64)
    for ly in range(nl):
        w = np.ones(p.size, np.float32)
        if ly % 2:
            w[1::2] = -1
        out[ly] = p * w
    return out


def test_port_estimator_with_caller_pilots_and_hopping(ctx):
    """miphy_port_channel_estimate_batch (port_channel_estimator::compute, port_channel_estimator.h:102-106). (1) With the pilots the PUSCH
    estimator would generate and no hopping it must reproduce miphy_dmrs_pusch_estimate_batch / the oracle. (2) With intra-slot hopping
    (port_channel_estimator_average_impl.cpp:97-146) every hop is an estimate of its own on its own PRBs: the oracle run per hop gives the
    coefficients of the hop's symbols; RSRP and EPRE are the DM-RS-symbol-weighted means of the hops, the noise variance is forced to
    EPRE / 1000 (:118-138), the time alignment is the mean of the hops'. Same tolerances as above."""
    import torch
    import miphy
    rng = np.random.default_rng(2718)
    for nl, hop, alloc1, alloc2, dsyms, delay in ((1, 0, slice(4, 34), None, [2, 11], 6.0), (2, 0, slice(0, 52), None, [3], -4.0),
                                                  (1, 7, slice(2, 22), slice(28, 48), [2, 9], 5.0), (2, 6, slice(0, 25), slice(27, 52), [2, 4, 8, 11], -9.0)):
        nprb, slot, scr, scaling = 52, 5, 321, 1.4125
        first, nof = (0, 14) if nl == 1 else (1, 12)
        dsyms = [l for l in dsyms if first <= l < first + nof]
        hops = [(first, first + nof, alloc1)] if not hop else [(first, hop, alloc1), (hop, first + nof, alloc2)]
        # the grid: every hop carries its own valid DM-RS (make_case on the hop's PRBs and symbols), added up
        g = np.zeros((1, 14, nprb * 12), np.complex64)
        per_hop = []
        for (a, b, alloc) in hops:
            ds = [l for l in dsyms if a <= l < b]
            case = make_case(rng, nprb, alloc, 1, nl, ds, slot=slot, scr=scr, scaling=scaling, delay=delay)
            gh = case[-1]
            g[:, a:b, :] = gh[:, a:b, :]
            per_hop.append((a, b, ds, case[7]))
        # expected per hop from the oracle on the hop's own allocation
        exp = []
        for (a, b, ds, rb) in per_hop:
            sm = np.zeros(14, np.uint8)
            sm[ds] = 1
            ce_h, sc_h = o_dmrs_pusch_estimate(1, slot, False, scr, 0, scaling, sm, rb, a, b - a, nl, g)
            exp.append((ce_h, sc_h))
        j = np.zeros(1, dtype=miphy.PuschChestJob)
        j[0]["numerology"], j[0]["scaling"], j[0]["nof_tx_layers"], j[0]["nof_rx_ports"] = 1, scaling, nl, 1
        j[0]["first_symbol"], j[0]["nof_symbols"], j[0]["rx_ports"], j[0]["grid_nof_prb"] = first, nof, [0, 1, 2, 3], nprb
        j[0]["symbols_mask"] = sum(1 << l for l in dsyms)
        for key, (a, b, ds, rb) in zip(("rb_mask", "rb_mask2"), per_hop):
            m = [0] * 5
            for r in np.nonzero(rb)[0]:
                m[r >> 6] |= 1 << (int(r) & 63)
            j[0][key] = m
        j[0]["hop_symbol"] = hop
        j[0]["re_odd_mask"] = sum(((ly // 2) % 2) << ly for ly in range(nl))
        # pilots: [layer][DM-RS symbol of the allocation, hop 1 first][pilot]
        pil = np.concatenate([np.stack([_pilots(rb, l, slot, scr, 0, nl) for l in ds], axis=1) for (a, b, ds, rb) in per_hop], axis=1)
        nsymb = first + nof
        g_d, p_d = torch.from_numpy(g.reshape(-1)).cuda(), torch.from_numpy(np.ascontiguousarray(pil).reshape(-1)).cuda()
        ce_d = torch.ones(nl * nsymb * nprb * 12, dtype=torch.complex64, device="cuda")
        sc_d = torch.zeros(5 * nl, dtype=torch.float32, device="cuda")
        ctx.port_channel_estimate_batch(j, g_d, p_d, ce_d, sc_d)
        torch.cuda.synchronize()
        ce, sc = ce_d.cpu().numpy().reshape(nl, 1, nsymb, nprb * 12), sc_d.cpu().numpy().reshape(1, nl, 5)
        nds_all = len(dsyms)
        for h, ((a, b, ds, rb), (ce_h, sc_h)) in enumerate(zip(per_hop, exp)):
            mask = np.repeat(rb.astype(bool), 12)
            got, want = ce[:, :, a:b][..., mask], ce_h[:, :, a:b][..., mask]
            assert np.abs(got - want).max() / np.abs(want).max() < 1e-4, (nl, hop, h)
            assert np.all(ce[:, :, a:b][..., ~mask] == 1.0), "only the hop's PRBs are written in the hop's symbols"
        w = np.array([len(ds) for (_, _, ds, _) in per_hop], np.float64) / nds_all
        rsrp = sum(wi * e[1][..., 0].astype(np.float64) for wi, e in zip(w, exp))
        epre = sum(wi * e[1][..., 1].astype(np.float64) for wi, e in zip(w, exp))
        assert np.all(np.abs(sc[..., 0] - rsrp) <= 1e-4 * rsrp) and np.all(np.abs(sc[..., 1] - epre) <= 1e-4 * epre), (nl, hop)
        if hop:
            assert np.all(np.abs(sc[..., 2] - 0.001 * epre) <= 1e-4 * 0.001 * epre)
            assert np.all(np.abs(sc[..., 3] - rsrp / scaling ** 2 / (0.001 * epre)) <= 2e-4 * sc[..., 3])
            ta = sum(e[1][..., 4].astype(np.float64) for e in exp) / 2
        else:
            for k in (2, 3):
                assert np.all(np.abs(sc[..., k] - exp[0][1][..., k]) <= 1e-4 * np.abs(exp[0][1][..., k]))
            ta = exp[0][1][..., 4].astype(np.float64)
        assert np.abs(sc[..., 4] - ta).max() <= 1.01 / (4096 * 30000.0), (nl, hop, sc[..., 4], ta)
