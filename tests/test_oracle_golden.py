"""CPU: the oracle (oracle/phy_oracle.c) against the golden fixtures that oracle/gen_golden.py produced BY RUNNING THE
REFERENCE (srsRAN_Project 23.5, AVX2 paths). Integer work bit-exact; floating point within the stated tolerances.
This is what pins the oracle on machines where /root/reference does not exist (the GPU box)."""
import os

import numpy as np
import pytest

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def count(d, prefix):
    return len([k for k in d.files if k.startswith(prefix)])


def test_crc():
    d = load("crc")
    for i in range(count(d, "bits_")):
        poly, exp = d["meta_%d" % i]
        assert O.o_crc_bits(int(poly), d["bits_%d" % i]) == int(exp)


def test_ldpc_encoder_decoder():
    d = load("ldpc_enc_dec")
    for i in range(count(d, "msg_")):
        bg, Z, nf, L = (int(x) for x in d["meta_%d" % i])
        cw = O.o_ldpc_encode(bg, Z, d["msg_%d" % i], O.BG_NS[bg] * Z)
        assert np.array_equal(cw, d["cw_%d" % i]), (bg, Z)
        for row in d["dec_%d" % i]:
            crc, mi, it = (int(x) for x in row[:3])
            ito, bits = O.o_ldpc_decode(bg, Z, d["llr_%d" % i], nf, crc, mi)
            assert ito == it and np.array_equal(bits, row[3:].astype(np.uint8)), (bg, Z, crc, mi)


def test_rate_matcher_dematcher():
    d = load("ldpc_rate_match")
    for i in range(count(d, "cb_")):
        bg, Z, rv, mod, Nref, nf, E = (int(x) for x in d["meta_%d" % i])
        assert np.array_equal(O.o_rate_match(rv, mod, Nref, nf, d["cb_%d" % i], E), d["rm_%d" % i])
        assert np.array_equal(O.o_rate_dematch(rv, mod, Nref, nf, 1, d["llr_%d" % i], d["sb_%d" % i]), d["rdm_new_%d" % i])
        assert np.array_equal(O.o_rate_dematch(rv, mod, Nref, nf, 0, d["llr_%d" % i], d["sb_%d" % i]), d["rdm_comb_%d" % i])


def test_sch_chain_with_harq():
    d = load("sch_chain")
    for i in range(count(d, "tb_")):
        bg, mod, nl, nsym, tbs = (int(x) for x in d["meta_%d" % i])
        tb = d["tb_%d" % i]
        rvs = [0, 2, 3, 1]
        for t, rv in enumerate(rvs):
            assert np.array_equal(O.o_pdsch_encode(bg, rv, mod, 0, nl, nsym, tb), d["cw_%d" % i][t])
        od = O.OraclePuschDecoder(bg, mod, 0, nl, nsym, tbs // 8)
        for t, rv in enumerate(rvs):
            ok, tbo, mm = od.decode(d["llr_%d" % i][t], rv, t == 0, 6, True)
            eo, ea, eb = (int(x) for x in d["res_%d" % i][t])
            assert (int(ok), mm[0], mm[1]) == (eo, ea, eb), (i, t)
            if ok:
                assert np.array_equal(tbo, d["tbo_%d" % i][t])


def test_dft():
    """Reference = float radix-2 generic DFT; oracle = exact transform. Tolerance 3e-6 * rms (the reference's own test allows
    MSE < 1e-6, dft_processor_test.cpp:40-42)."""
    d = load("dft")
    for i in range(count(d, "x_")):
        for key, inv in (("fwd", False), ("inv", True)):
            ref = d["%s_%d" % (key, i)]
            got = O.o_dft(d["x_%d" % i], inv)
            assert np.abs(got - ref).max() < 3e-6 * np.sqrt(np.mean(np.abs(ref) ** 2))


def test_ofdm():
    d = load("ofdm")
    for i in range(count(d, "x_")):
        mu, rb, N, wo, fc, slot = d["meta_%d" % i]
        cfg = O.OfdmCfg(int(mu), int(rb), int(N), int(wo), 0.5, float(fc))
        g = O.o_ofdm_demod_slot(cfg, int(slot), d["x_%d" % i])
        ref = d["grid_%d" % i]
        assert np.abs(g - ref).max() < 3e-6 * np.sqrt(np.mean(np.abs(ref) ** 2))
        y = O.o_ofdm_mod_slot(O.OfdmCfg(int(mu), int(rb), int(N), 0, 0.01, float(fc)), int(slot), d["g_%d" % i])
        ref = d["y_%d" % i]
        assert np.abs(y - ref).max() < 3e-6 * np.sqrt(np.mean(np.abs(ref) ** 2))


def test_dmrs_pusch_estimator():
    """Tolerances: coefficients 1e-5 * max|h|, scalars 1e-5 relative, time alignment exact (reference test: 5e-4)."""
    d = load("dmrs_pusch_estimator")
    for i in range(count(d, "grid_")):
        mu, slot, scr, nscid, scaling, nl = d["meta_%d" % i]
        rb, sm = d["rb_%d" % i], d["sm_%d" % i]
        ce, sc = O.o_dmrs_pusch_estimate(int(mu), int(slot), False, int(scr), int(nscid), float(scaling), sm, rb, 0, 14, int(nl), d["grid_%d" % i])
        mask = np.repeat(rb.astype(bool), 12)
        ref = d["ce_%d" % i]
        assert np.abs(ce[..., mask] - ref).max() < 1e-5 * np.abs(ref).max()
        rsc = d["sc_%d" % i]
        assert np.all(np.abs(sc[..., :4] - rsc[..., :4]) <= 1e-5 * np.abs(rsc[..., :4]))
        assert np.array_equal(sc[..., 4], rsc[..., 4])


def test_polar_and_pdcch():
    d = load("polar")
    for i in range(count(d, "msg_")):
        K, E, nMax, ibil = (int(x) for x in d["meta_%d" % i])
        rm, al, en = O.o_polar_encode_chain(K, E, nMax, ibil, d["msg_%d" % i])
        assert np.array_equal(rm, d["rm_%d" % i]) and np.array_equal(al, d["alloc_%d" % i]) and np.array_equal(en, d["enc_%d" % i])
        m, dem, u = O.o_polar_decode_chain(K, E, nMax, ibil, d["llr_%d" % i])
        assert np.array_equal(m, d["dec_msg_%d" % i]) and np.array_equal(dem, d["dem_%d" % i]) and np.array_equal(u, d["u_%d" % i])
    for i in range(count(d, "pdcch_pay_")):
        A, E, rnti = (int(x) for x in d["pdcch_meta_%d" % i])
        assert np.array_equal(O.o_pdcch_encode(d["pdcch_pay_%d" % i], rnti, E), d["pdcch_out_%d" % i])


def test_pbch():
    d = load("polar")
    for i in range(count(d, "pbch_pay_")):
        a = [int(x) for x in d["pbch_meta_%d" % i]]
        assert np.array_equal(O.o_pbch_encode(*a, d["pbch_pay_%d" % i]), d["pbch_out_%d" % i])


def test_llr_algebra_kats():
    """Known answers of tests/unittests/phy/upper/log_likelihood_ratio_test.cpp:37-86 as they surface through the oracle's
    rate-dematcher combine (clamp to +-120) and the polar repetition combine (promotion to +-127)."""
    # saturating combine: 100 + 50 -> 120, -100 + -50 -> -120, 5 + -5 -> 0
    N = 66 * 2
    sb = np.zeros(N, np.int8)
    sb[:3] = [100, -100, 5]
    llr = np.array([50, -50, -5, 0], np.int8)
    out = O.o_rate_dematch(0, 1, 0, 0, 0, llr, sb)
    assert list(out[:3]) == [120, -120, 0]
    # promotion sum through polar repetition (E >= N): 100 + 50 -> +inf (127)
    K, E = 36, 1728  # PDCCH AL16: N = 512 < E, every codeword position receives 3 or 4 repetitions
    llr = np.zeros(E, np.int8)
    llr[0], llr[512] = 100, 50
    llr[1], llr[513] = -100, -50
    llr[2], llr[514] = 127, -127
    _, dem, _ = O.o_polar_decode_chain(K, E, 9, 0, llr)
    assert dem.size == 512
    # the sub-block interleaver maps positions 0, 1, 2 to themselves (P(0) = 0)
    assert dem[0] == 127 and dem[1] == -127 and dem[2] == 0


def test_pusch_demodulator():
    """Soft demapper and whole PUSCH demodulator against reference-produced LLRs. Stated tolerance: at most one quantisation step
    (the reference equalises with the approximate _mm256_rcp_ps and its build may contract a*b+c), and at least 99 % of the LLRs
    identical (97 % at the high SNR of these stimuli, where the LLR magnitudes are large); the demapper alone (same inputs, exact arithmetic on both sides) must agree on > 99.99 %."""
    d = load("pusch_demod")
    for i in range(count(d, "dm_sym_")):
        mod = int(d["dm_meta_%d" % i][0])
        got, ref = O.o_demodulate_soft(mod, d["dm_sym_%d" % i], d["dm_nv_%d" % i]), d["dm_llr_%d" % i]
        diff = np.abs(got.astype(int) - ref.astype(int))
        assert diff.max() <= 1 and (diff == 0).mean() > 0.9999, (mod, diff.max(), (diff == 0).mean())
    for i in range(count(d, "grid_")):
        rnti, n_id, mod, start, nof, cdm, nv = d["meta_%d" % i]
        got, eq, nvar = O.o_pusch_demodulate(int(rnti), int(n_id), int(mod), int(start), int(nof), d["dm_%d" % i], 0, int(cdm), d["rb_%d" % i],
                                             d["grid_%d" % i], d["ce_%d" % i], float(nv))
        ref = d["llr_%d" % i]
        diff = np.abs(got.astype(int) - ref.astype(int))
        assert diff.max() <= 1 and (diff == 0).mean() > 0.97, (i, diff.max(), (diff == 0).mean())
        # the stimulus carries known (unscrambled) bits at high SNR: hard decisions of the descrambled LLRs = bits xor c(n)
        bits = d["bits_%d" % i] ^ O.o_gold((int(rnti) << 15) + int(n_id), 0, got.size)
        nz = got != 0
        assert ((got < 0).astype(np.uint8)[nz] == bits[nz]).mean() > (0.999 if mod <= 4 else 0.9)  # 64/256QAM: the stimulus noise flips some LSBs


def test_channel_equalizer():
    """Stand-alone zero-forcing equalizer against reference-produced outputs (create_channel_equalizer_factory_zf). Stated tolerance:
    one layer -- the reference's AVX2 path divides with the approximate _mm256_rcp_ps (relative error <= 1.5 * 2^-12): 4e-4 relative;
    two layers on two ports -- scalar code on both sides, the reference build may contract a*b+c into fused multiply-adds: 2e-5 of
    the largest magnitude. Dead estimates give symbol 0 and noise variance +inf on both sides."""
    d = load("channel_equalizer")
    for i in range(int(d["n"])):
        y, h, zr, nvr = d["y_%d" % i], d["h_%d" % i], d["z_%d" % i], d["nv_%d" % i]
        nvar, txs = (float(v) for v in d["meta_%d" % i])
        z, nv = O.o_channel_equalize(y, h, nvar, txs)
        assert np.array_equal(np.isinf(nv), np.isinf(nvr)) and np.isinf(nv).sum() == 2 * h.shape[0]
        assert np.all(z[np.isinf(nv)] == 0) and np.all(zr[np.isinf(nvr)] == 0)
        fin = ~np.isinf(nv)
        if h.shape[0] == 1:
            assert np.all(np.abs(z[fin] - zr[fin]) <= 4e-4 * np.abs(zr[fin]) + 1e-7), i
            assert np.all(np.abs(nv[fin] - nvr[fin]) <= 4e-4 * nvr[fin]), i
        else:
            # the 2 x 2 denominator n0 n1 - |xi|^2 cancels: compare in units of what went into the subtraction
            n0 = (np.abs(h[0]) ** 2).sum(0)
            n1 = (np.abs(h[1]) ** 2).sum(0)
            cond = (n0 * n1) / np.maximum(n0 * n1 - np.abs((h[0].conj() * h[1]).sum(0)) ** 2, 1e-30)
            tol = 2e-6 * cond
            for l in range(2):
                f = fin[l]
                assert np.all(np.abs(z[l][f] - zr[l][f]) <= tol[f] * (np.abs(zr[l][f]) + 1.0)), i
                assert np.all(np.abs(nv[l][f] - nvr[l][f]) <= tol[f] * nvr[l][f]), i
    assert O.o_channel_equalize(np.zeros((3, 4), np.complex64), np.zeros((2, 3, 4), np.complex64), 0.1, 1.0) is None  # 2 layers on 3 ports


def test_pdsch_modulator_and_dmrs():
    """Modulation mapper, PDSCH modulator (scrambling, mapping around DM-RS and reserved patterns, scaling incl. NaN = none) and PDSCH
    DM-RS mapping against reference-produced grids: bit-exact single-precision values."""
    d = load("pdsch_mod")
    for i in range(count(d, "map_bits_")):
        got = O.o_modulate(int(d["map_meta_%d" % i][0]), d["map_bits_%d" % i])
        assert np.array_equal(got.view(np.uint32), d["map_sym_%d" % i].view(np.uint32))
    for i in range(count(d, "pm_cw_")):
        rnti, n_id, scaling, mod, start, nof, type2, cdm, bwp_start, bwp_size, port, nprb_grid, ngp = d["pm_meta_%d" % i]
        ref = d["pm_grid_%d" % i]
        res = [(d["pm_res_prb_%d" % i][r], int(d["pm_res_re_%d" % i][r, 0]), int(d["pm_res_re_%d" % i][r, 1])) for r in range(d["pm_res_prb_%d" % i].shape[0])]
        g = np.zeros_like(ref)
        n = O.o_pdsch_modulate(int(rnti), int(n_id), float(scaling), 1, [int(mod)], [d["pm_cw_%d" % i]], int(start), int(nof), d["pm_dm_%d" % i], int(type2),
                               int(cdm), int(bwp_start), int(bwp_size), d["pm_prb_%d" % i], res, [int(port)], int(nprb_grid), g)
        assert n * int(mod) == d["pm_cw_%d" % i].size
        assert np.array_equal(g.view(np.uint32), ref.view(np.uint32)), i
    for i in range(count(d, "dd_grid_")):
        slot, ref_pt, type2, scr, nscid, amp, nports = d["dd_meta_%d" % i]
        ref = d["dd_grid_%d" % i]
        g = np.zeros_like(ref)
        O.o_dmrs_pdsch_map(int(slot), int(ref_pt), int(type2), int(scr), int(nscid), float(amp), d["dd_sm_%d" % i], d["dd_rb_%d" % i], list(range(int(nports))), g)
        assert np.array_equal(g.view(np.uint32), ref.view(np.uint32)), i


def test_ofh_iq_compression():
    """Open Fronthaul IQ formats (BFP and uncompressed): oracle against payloads / samples recorded from the reference's avx2 and
    generic classes."""
    g = np.load(os.path.join(GOLD, "ofh_iq.npz"))
    for i, (comp, w, nprb) in enumerate(g["cases"].tolist()):
        pl = g["dec_payload_%d" % i]
        assert np.array_equal(O.o_ofh_iq_decompress(pl, nprb, w, True, comp).view(np.uint32), g["dec_simd_%d" % i].view(np.uint32)), (comp, w, nprb)
        assert np.array_equal(O.o_ofh_iq_decompress(pl, nprb, w, False, comp).view(np.uint32), g["dec_generic_%d" % i].view(np.uint32)), (comp, w, nprb)
        if w >= 8:
            for sc in (1.0, 0.37):
                assert np.array_equal(O.o_ofh_iq_compress(g["cmp_in_%d" % i], nprb, w, sc, comp), g["cmp_out_%d_%d" % (i, int(sc * 100))]), (comp, w, nprb, sc)
    # The product form of the SIMD classes and the division of the generic class give the same single-precision value for every
    # 9-bit sample and exponent 0..7 (checked exhaustively below), so the recorded outputs of the two classes coincide.
    for i in range(len(g["cases"])):
        assert np.array_equal(g["dec_simd_%d" % i].view(np.uint32), g["dec_generic_%d" % i].view(np.uint32))
    x = np.arange(-256, 256, dtype=np.int32)
    for e in range(8):
        r = np.float32(1.0) / (np.float32(32767.0) / np.float32(1 << e))
        assert np.array_equal((x.astype(np.float32) * r).view(np.uint32), ((x << e).astype(np.float32) / np.float32(32767.0)).view(np.uint32))


def test_pdcch_processor():
    """PDCCH processor: oracle against grids recorded from the reference processor (all three CCE-to-REG mapping types)."""
    g = np.load(os.path.join(GOLD, "pdcch_proc.npz"))
    kinds = set()
    for i in range(int(g["n"])):
        slot, rnti, nd, nr, ndm, ref, xdb, ddb, AL, start, dur, mapping = g["meta_%d" % i]
        out = np.zeros_like(g["grid_%d" % i])
        n = O.o_pdcch_process(int(slot), int(rnti), int(nd), int(nr), int(ndm), int(ref), float(xdb), float(ddb), g["pay_%d" % i], int(AL), int(start), int(dur),
                              g["rb_%d" % i], out)
        assert n == 54 * int(AL) and np.array_equal(out.view(np.uint32), g["grid_%d" % i].view(np.uint32)), i
        kinds.add(int(mapping))
    assert kinds == {0, 1, 2}


def test_ssb_processor():
    """SS/PBCH block: oracle against grids recorded from the reference processor (PBCH, its DM-RS, PSS, SSS; bit patterns incl. signed zeros)."""
    g = np.load(os.path.join(GOLD, "ssb_proc.npz"))
    for i in range(int(g["n"])):
        N_id, ssb_idx, L_max, hrf, sfn, kssb, k0, l0, beta, case = g["meta_%d" % i]
        out = np.zeros_like(g["grid_%d" % i])
        assert O.o_ssb_process(int(N_id), int(ssb_idx), int(L_max), int(hrf), int(sfn), int(kssb), g["pay_%d" % i], int(k0), int(l0), float(beta), 106, out) == 0
        assert np.array_equal(out.view(np.uint32), g["grid_%d" % i].view(np.uint32)), i


def test_nzp_csi_rs_generator():
    """NZP-CSI-RS: oracle against grids recorded from the reference generator (mapping rows 1-8, every density and CDM type it supports)."""
    g = np.load(os.path.join(GOLD, "csi_rs.npz"))
    rows = set()
    for i in range(int(g["n"])):
        slot, scr, amp, start_rb, nof_rb, b, e, st, row, cdm, dens, nports = g["meta_%d" % i]
        out = np.zeros_like(g["grid_%d" % i])
        assert O.o_csi_rs_map(int(slot), int(scr), float(amp), int(start_rb), int(nof_rb), (int(b), int(e), int(st)), int(row), int(cdm), int(dens),
                              list(range(int(nports))), g["rm_%d" % i], g["sm_%d" % i], 80, out) == 0
        assert np.array_equal(out.view(np.uint32), g["grid_%d" % i].view(np.uint32)), i
        rows.add(int(row))
    assert rows == {1, 2, 3, 4, 5, 6, 7, 8}
