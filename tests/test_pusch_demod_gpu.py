"""GPU: PUSCH demodulator kernel (RE extraction + ZF/MRC equalisation + soft demapping + descrambling) through the
C ABI against the oracle (bit-exact: both sides use single IEEE operations in the same order) and against reference-produced
LLRs from tests/golden/pusch_demod.npz (stated tolerance: one quantisation step, see test_oracle_golden.py)."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _mask_words(rb):
    w = np.zeros(5, dtype=np.uint64)
    for r in np.nonzero(rb)[0]:
        w[r >> 6] |= np.uint64(1) << np.uint64(r & 63)
    return w


def _job(miphy, rnti, n_id, mod, start, nof, dm, type2, cdm, rb, ports, ce_syms, grid_off=0, ce_off=0, sc_off=0, llr_off=0):
    j = np.zeros(1, dtype=miphy.PuschDemodJob)[0]
    j["rnti"], j["n_id"], j["mod"], j["nof_rx_ports"], j["start_symbol"], j["nof_symbols"] = rnti, n_id, mod, ports, start, nof
    j["dmrs_type"], j["nof_cdm_groups_without_data"], j["ce_nof_symbols"] = 2 if type2 else 1, cdm, ce_syms
    j["rx_ports"] = [0, 1, 2, 3]
    j["dmrs_symbols_mask"] = sum(1 << int(s) for s in np.nonzero(dm)[0])
    j["grid_nof_prb"] = rb.size
    j["rb_mask"] = _mask_words(rb)
    j["grid_offset"], j["ce_offset"], j["scalars_offset"], j["llr_offset"] = grid_off, ce_off, sc_off, llr_off
    j["nof_llr"] = miphy.pusch_demod_nof_llr(j)
    return j


def _run(ctx, jobs, grids, ces, nvs, total_llr):
    import torch
    import miphy
    g = torch.from_numpy(np.concatenate([x.reshape(-1) for x in grids])).cuda()
    h = torch.from_numpy(np.concatenate([x.reshape(-1) for x in ces])).cuda()
    sc = np.zeros(5 * len(nvs), dtype=np.float32)
    sc[2::5] = nvs
    out = torch.full((total_llr + 64,), 99, dtype=torch.int8, device="cuda")
    ctx.pusch_demodulate_batch(np.array(jobs, dtype=miphy.PuschDemodJob), g, h, torch.from_numpy(sc).cuda(), out)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_golden_vectors_and_oracle(ctx):
    import miphy
    d = np.load(os.path.join(GOLD, "pusch_demod.npz"))
    n = sum(1 for k in d.files if k.startswith("grid_"))
    jobs, grids, ces, nvs, offs, exp = [], [], [], [], [], []
    goff = coff = loff = 0
    for i in range(n):
        rnti, n_id, mod, start, nof, cdm, nv = d["meta_%d" % i]
        grid, ce, rb, dm = d["grid_%d" % i], d["ce_%d" % i], d["rb_%d" % i], d["dm_%d" % i]
        j = _job(miphy, int(rnti), int(n_id), int(mod), int(start), int(nof), dm, 0, int(cdm), rb, grid.shape[0], 14, goff, coff, 5 * i, loff)
        assert j["nof_llr"] == d["llr_%d" % i].size
        jobs.append(j)
        grids.append(grid)
        ces.append(ce)
        nvs.append(nv)
        offs.append((loff, int(j["nof_llr"])))
        exp.append(O.o_pusch_demodulate(int(rnti), int(n_id), int(mod), int(start), int(nof), dm, 0, int(cdm), rb, grid, ce, float(nv))[0])
        goff += grid.size
        coff += ce.size
        loff += int(j["nof_llr"]) + 3  # deliberately unaligned codeword starts
    out = _run(ctx, jobs, grids, ces, nvs, loff)
    for i, (o, ln) in enumerate(offs):
        got = out[o:o + ln]
        assert np.array_equal(got, exp[i]), (i, int(np.abs(got.astype(int) - exp[i].astype(int)).max()))
        ref = d["llr_%d" % i]
        diff = np.abs(got.astype(int) - ref.astype(int))
        assert diff.max() <= 1 and (diff == 0).mean() > 0.97
        assert np.all(out[o + ln:o + ln + 3] == 99)  # nothing written past the codeword


@pytest.mark.parametrize("mod,ports,cdm,type2,nprb,start,nof,dsyms", [
    (8, 1, 2, 0, 273, 0, 14, (2,)),          # the 100 MHz workload of the benchmark
    (6, 4, 1, 0, 106, 0, 14, (2, 7, 11)),
    (4, 2, 2, 0, 52, 2, 12, (3, 10)),
    (2, 1, 1, 0, 25, 0, 14, (2, 11)),
    (1, 2, 2, 0, 11, 1, 9, (4,)),
    (6, 1, 1, 1, 40, 0, 14, (2,)),            # DM-RS type 2
    (8, 3, 2, 1, 33, 0, 13, (2, 11)),
    (4, 2, 3, 1, 20, 0, 14, (2,)),            # type 2, all CDM groups: no data on the DM-RS symbol
])
def test_random_allocations_match_oracle(ctx, mod, ports, cdm, type2, nprb, start, nof, dsyms):
    import miphy
    rng = np.random.default_rng(1000 * mod + nprb)
    nsc = nprb * 12
    rb = (rng.uniform(size=nprb) < 0.85).astype(np.uint8)
    rb[nprb // 2] = 1
    if nprb == 273:
        rb[:] = 1
    dm = np.zeros(14, np.uint8)
    dm[list(dsyms)] = 1
    grid = (rng.standard_normal((ports, 14, nsc)) + 1j * rng.standard_normal((ports, 14, nsc))).astype(np.complex64)
    ce = (rng.standard_normal((ports, 14, nsc)) + 1j * rng.standard_normal((ports, 14, nsc))).astype(np.complex64)
    ce[0, start, 7] = 0
    grid[0, start + 1, 3] = np.nan
    rnti, n_id, nv = int(rng.integers(1, 65536)), int(rng.integers(0, 1024)), float(rng.uniform(0.01, 0.5))
    exp, _, _ = O.o_pusch_demodulate(rnti, n_id, mod, start, nof, dm, type2, cdm, rb, grid, ce, nv)
    j = _job(miphy, rnti, n_id, mod, start, nof, dm, type2, cdm, rb, ports, 14)
    assert j["nof_llr"] == exp.size
    out = _run(ctx, [j], [grid], [ce], [nv], exp.size)
    assert np.array_equal(out[:exp.size], exp)


@pytest.mark.parametrize("device_jobs", [False, True])
@pytest.mark.parametrize("mod,ports,cdm,type2,nprb,start,nof,dsyms,pad", [
    (8, 1, 2, 0, 273, 0, 14, (2,), 0),      # the benchmark's slot: every request of a thread in flight at once (demod_columns_deep)
    (6, 1, 2, 0, 106, 0, 14, (2, 7, 11), 0),
    (4, 1, 1, 0, 52, 2, 12, (3, 10), 0),
    (2, 1, 1, 1, 25, 1, 9, (4,), 0),         # DM-RS type 2, a partial slot
    (8, 1, 2, 0, 40, 0, 14, (2,), 3),        # codeword at an odd address: the byte-wise store path of the same walk
    (4, 1, 3, 1, 20, 0, 14, (2,), 0),        # type 2, all CDM groups: no data on the DM-RS symbol
    (6, 2, 2, 0, 60, 0, 14, (2, 11), 0),     # two ports with the compact estimate: the general walk
    (1, 1, 2, 0, 30, 0, 14, (2,), 0),        # pi/2-BPSK: the general walk
])
def test_compact_estimate_matches_oracle(ctx, mod, ports, cdm, type2, nprb, start, nof, dsyms, pad, device_jobs):
    """The estimate as ONE row per port (ce_compact, what the estimator of this library hands over): the demodulator then keeps the channel
    row in registers, and with one port it requests the samples of all OFDM symbols at once. Same LLRs as the oracle fed with that row
    on every symbol, including a zero channel coefficient, a NaN sample and partial allocations; descriptors in host and in device memory."""
    import torch
    import miphy
    rng = np.random.default_rng(2000 * mod + nprb + pad)
    nsc = nprb * 12
    rb = (rng.uniform(size=nprb) < 0.85).astype(np.uint8)
    rb[nprb // 2] = 1
    if nprb == 273:
        rb[:] = 1
    dm = np.zeros(14, np.uint8)
    dm[list(dsyms)] = 1
    grid = (rng.standard_normal((ports, 14, nsc)) + 1j * rng.standard_normal((ports, 14, nsc))).astype(np.complex64)
    row = (rng.standard_normal((ports, 1, nsc)) + 1j * rng.standard_normal((ports, 1, nsc))).astype(np.complex64)
    row[0, 0, 12 * (nprb // 2) + 5] = 0
    grid[0, start + 1, 12 * (nprb // 2) + 3] = np.nan
    rnti, n_id, nv = int(rng.integers(1, 65536)), int(rng.integers(0, 1024)), float(rng.uniform(0.01, 0.5))
    exp, _, _ = O.o_pusch_demodulate(rnti, n_id, mod, start, nof, dm, type2, cdm, rb, grid, np.repeat(row, 14, axis=1), nv)
    j = _job(miphy, rnti, n_id, mod, start, nof, dm, type2, cdm, rb, ports, 14, llr_off=pad)
    j["ce_compact"] = 1
    assert j["nof_llr"] == exp.size
    g = torch.from_numpy(grid.reshape(-1)).cuda()
    h = torch.from_numpy(row.reshape(-1)).cuda()
    sc = np.zeros(5, dtype=np.float32)
    sc[2] = nv
    out = torch.full((exp.size + pad + 64,), 99, dtype=torch.int8, device="cuda")
    jobs = np.array([j], dtype=miphy.PuschDemodJob)
    ctx.pusch_demodulate_batch(torch.from_numpy(jobs.view(np.uint8)).cuda() if device_jobs else jobs, g, h, torch.from_numpy(sc).cuda(), out)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got[pad:pad + exp.size], exp)
    assert np.all(got[:pad] == 99) and np.all(got[pad + exp.size:] == 99)


def test_rejections(ctx):
    """pusch_demodulator_impl.cpp:76-83 asserts the codeword length and the single layer; the C ABI reports MIPHY_EINVAL."""
    import torch
    import miphy
    rb = np.ones(10, np.uint8)
    dm = np.zeros(14, np.uint8)
    dm[2] = 1
    x = torch.zeros(4 * 14 * 120, dtype=torch.complex64, device="cuda")
    f = torch.zeros(16, dtype=torch.float32, device="cuda")
    o = torch.zeros(40000, dtype=torch.int8, device="cuda")
    for bad in (dict(nof_llr=8), dict(mod=3), dict(nof_rx_ports=5), dict(dmrs_type=3), dict(nof_cdm_groups_without_data=3), dict(nof_symbols=15),
                dict(ce_nof_symbols=5)):
        j = _job(miphy, 1, 2, 4, 0, 14, dm, 0, 2, rb, 1, 14)
        for k, v in bad.items():
            j[k] = v
        with pytest.raises(RuntimeError):
            ctx.pusch_demodulate_batch(np.array([j], dtype=miphy.PuschDemodJob), x, x, f, o)


@pytest.mark.parametrize("mod,ports,cdm", [(2, 1, 2), (4, 2, 2), (6, 1, 1), (8, 4, 2), (1, 1, 2)])
def test_placeholders_and_evm_match_oracle(ctx, mod, ports, cdm):
    """UCI on PUSCH: repetition placeholders in the descrambler (bit-exact LLRs against the oracle) and the per-symbol EVM sums
    (EVM within 2e-6 relative of the oracle's sequential sum: the kernel adds in a tree)."""
    import torch
    import miphy
    rng = np.random.default_rng(900 + mod)
    nprb = 31
    nsc = nprb * 12
    rb = np.zeros(nprb, np.uint8)
    rb[1:27] = 1
    dm = np.zeros(14, np.uint8)
    dm[[3, 10]] = 1
    grid = (rng.standard_normal((ports, 14, nsc)) + 1j * rng.standard_normal((ports, 14, nsc))).astype(np.complex64)
    ce = (rng.standard_normal((ports, 14, nsc)) + 1j * rng.standard_normal((ports, 14, nsc))).astype(np.complex64)
    n_re = O.pusch_nof_re(1, 12, dm, 0, cdm, rb)
    ph = np.sort(rng.choice(n_re, 53, replace=False)).astype(np.uint16) if mod >= 2 else np.zeros(0, np.uint16)
    j = _job(miphy, 0x1234, 77, mod, 1, 12, dm, 0, cdm, rb, ports, 14)
    j["placeholders_offset"], j["nof_placeholders"], j["evm_offset"] = 5, ph.size, 3
    g = torch.from_numpy(grid.reshape(-1)).cuda()
    h = torch.from_numpy(ce.reshape(-1)).cuda()
    sc = np.zeros(5, np.float32)
    sc[2] = 0.07
    out = torch.full((int(j["nof_llr"]) + 64,), 99, dtype=torch.int8, device="cuda")
    ph_d = torch.from_numpy(np.concatenate([np.zeros(5, np.uint16), ph, np.zeros(3, np.uint16)]).view(np.int16)).cuda()
    evm_d = torch.full((3 + 14 + 2,), -1.0, dtype=torch.float32, device="cuda")
    ctx.pusch_demodulate_batch_ex(np.array([j], dtype=miphy.PuschDemodJob), g, h, torch.from_numpy(sc).cuda(), out, ph_d, evm_d)
    torch.cuda.synchronize()
    o, oe = O.o_pusch_demodulate_ex(0x1234, 77, mod, 1, 12, dm, 0, cdm, rb, grid, ce, 0.07, placeholders=ph)
    got = out.cpu().numpy()
    assert np.array_equal(got[:o.size], o) and np.all(got[o.size:] == 99)
    e = evm_d.cpu().numpy()
    assert np.all(e[:3] == -1.0) and np.all(e[17:] == -1.0) and e[3] == 0.0 and e[3 + 13] == 0.0  # symbols outside the allocation contribute 0
    evm = float(np.sqrt(e[3:17].astype(np.float64).sum() / n_re))
    assert abs(evm - oe) <= 2e-6 * oe + 1e-7, (evm, oe)
