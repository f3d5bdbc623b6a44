"""CPU: the oracle against the reference itself (oracle/_ref/libref_capi.so, built from /root/reference by
oracle/build_ref.sh). Skipped where the reference build is not present. Seeded, sized to run in well under a minute."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (no /root/reference here)")


def noisy(cw, sigma, rng):
    y = (1.0 - 2.0 * (cw & 1)) + sigma * rng.standard_normal(cw.size)
    return np.round(np.clip(4 * y, -20, 20) / 20 * 120).astype(np.int8)


def test_crc_all_polynomials():
    rng = np.random.default_rng(0)
    for poly in range(5):  # CRC6 is not on the hot path (UCI only) and not restated
        for n in (1, 7, 8, 9, 24, 100, 1001, 8424, 8423):
            b = rng.integers(0, 2, n, dtype=np.uint8)
            assert O.o_crc_bits(poly, b) == O.r_crc_bits(poly, b) == O.r_crc_bits(poly, b, True)


def test_ldpc_encoder_all_102_graphs():
    rng = np.random.default_rng(1)
    for bg in (1, 2):
        for Z in O.ALL_Z:
            K = O.BG_K[bg] * Z
            msg = rng.integers(0, 2, K, dtype=np.uint8)
            nf = int(rng.integers(0, Z))
            if nf:
                msg[-nf:] = 254
            for ol in (O.BG_NS[bg] * Z, K + 2 * Z):
                o = O.o_ldpc_encode(bg, Z, msg, ol)
                assert np.array_equal(o, O.r_ldpc_encode(bg, Z, msg, ol, "avx2"))
                assert np.array_equal(o, O.r_ldpc_encode(bg, Z, msg, ol, "generic"))


def test_ldpc_decoder_all_102_graphs_avx2():
    rng = np.random.default_rng(2)
    dec = O.RefLdpcDecoder("avx2")
    for bg in (1, 2):
        for Z in O.ALL_Z:
            K, NS = O.BG_K[bg] * Z, O.BG_NS[bg] * Z
            msg = rng.integers(0, 2, K, dtype=np.uint8)
            poly, nb = (O.CRC24B, 24) if K > 60 else (O.CRC16, 16)
            c = O.o_crc_bits(poly, msg[:K - nb])
            msg[K - nb:] = [(c >> (nb - 1 - i)) & 1 for i in range(nb)]
            cw = O.o_ldpc_encode(bg, Z, msg, NS)
            for L, sigma in ((NS, 0.6), (K + 2 * Z, 0.3)):
                llr = noisy(cw[:L], sigma, rng)
                for crc, mi in ((poly, 6), (-1, 3)):
                    a = O.o_ldpc_decode(bg, Z, llr, 0, crc, mi)
                    b = dec.decode(bg, Z, llr, 0, crc, mi)
                    assert a[0] == b[0] and np.array_equal(a[1], b[1]), (bg, Z, L, crc, mi)
    # full-range inputs incl. infinities and zero tails
    for bg, Z in ((1, 384), (2, 8), (1, 3), (2, 208)):
        r = rng.integers(-127, 128, O.BG_NS[bg] * Z).astype(np.int8)
        r[-(Z + Z // 2):] = 0
        a, b = O.o_ldpc_decode(bg, Z, r, 0, -1, 4), dec.decode(bg, Z, r, 0, -1, 4)
        assert a[0] == b[0] and np.array_equal(a[1], b[1])
        z = np.zeros_like(r)
        a, b = O.o_ldpc_decode(bg, Z, z, 0, -1, 4), dec.decode(bg, Z, z, 0, -1, 4)
        assert a[0] == b[0] == 0 and np.array_equal(a[1], b[1])


def test_rate_match_dematch():
    rng = np.random.default_rng(3)
    for bg in (1, 2):
        for Z in (2, 7, 52, 384):
            N, K = O.BG_NS[bg] * Z, O.BG_K[bg] * Z
            for rv in range(4):
                for mod in (1, 2, 4, 6, 8):
                    for Nref in (0, N - 3 * Z):
                        nf = int(rng.integers(0, Z))
                        cb = rng.integers(0, 2, N, dtype=np.uint8)
                        if nf:
                            cb[K - 2 * Z - nf:K - 2 * Z] = 254
                        E = mod * int(rng.integers(1, 3 * N // mod))
                        assert np.array_equal(O.o_rate_match(rv, mod, Nref, nf, cb, E), O.r_rate_match(bg, Z, rv, mod, Nref, nf, cb, E))
                        llr = rng.integers(-120, 121, E).astype(np.int8)
                        sb = rng.integers(-120, 121, N).astype(np.int8)
                        for nd in (1, 0):
                            o = O.o_rate_dematch(rv, mod, Nref, nf, nd, llr, sb)
                            assert np.array_equal(o, O.r_rate_dematch(bg, Z, rv, mod, Nref, nf, nd, llr, sb, "avx2"))
                            assert np.array_equal(o, O.r_rate_dematch(bg, Z, rv, mod, Nref, nf, nd, llr, sb, "generic"))


def test_sch_chain_harq():
    rng = np.random.default_rng(4)
    for bg, mod, nl, nprb, tbs, sigma in ((2, 2, 1, 106, 3848, 1.3), (1, 4, 1, 106, 42016, 0.62), (1, 8, 1, 273, 319784, 0.45)):
        nsym = nprb * 156 * nl
        tb = rng.integers(0, 256, tbs // 8, dtype=np.uint8)
        rvs = [0, 2, 3, 1]
        cws = [O.o_pdsch_encode(bg, rv, mod, 0, nl, nsym, tb) for rv in rvs]
        for rv, cw in zip(rvs, cws):
            assert np.array_equal(cw, O.r_pdsch_encode(bg, rv, mod, 0, nl, nsym, tb, "avx2"))
        llrs = np.stack([noisy(c, sigma, rng) for c in cws])
        ok_r, tb_r, mm_r = O.RefPuschDecoder("avx2").decode_sequence(bg, mod, 0, nl, nsym, tbs // 8, rvs, llrs, 6, True)
        od = O.OraclePuschDecoder(bg, mod, 0, nl, nsym, tbs // 8)
        for t, rv in enumerate(rvs):
            ok, tbo, mm = od.decode(llrs[t], rv, t == 0, 6, True)
            assert ok == ok_r[t] and mm == mm_r[t]
            if ok:
                assert np.array_equal(tbo, tb_r[t]) and np.array_equal(tbo, tb)


def test_dft_ofdm_tolerance():
    rng = np.random.default_rng(5)
    for N in (128, 384, 1024, 3072, 4096):
        x = (rng.uniform(-1, 1, N) + 1j * rng.uniform(-1, 1, N)).astype(np.complex64)
        for inv in (False, True):
            o, r = O.o_dft(x, inv), O.r_dft(x, inv)
            assert np.abs(o - r).max() < 3e-6 * np.sqrt(np.mean(np.abs(r) ** 2))
    for mu, rb, N, wo, fc, slot in ((1, 273, 4096, 144, 3.5e9, 0), (1, 106, 2048, 72, 3.5e9, 1)):
        cfg = O.OfdmCfg(mu, rb, N, wo, 0.5, fc)
        ns = O.o_ofdm_slot_size(cfg, slot)
        x = ((rng.standard_normal(ns) + 1j * rng.standard_normal(ns)) * 0.7).astype(np.complex64)
        o, r = O.o_ofdm_demod_slot(cfg, slot, x), O.r_ofdm_demod_slot(cfg, slot, x)
        assert np.abs(o - r).max() < 3e-6 * np.sqrt(np.mean(np.abs(r) ** 2))


def test_dmrs_pusch_estimator():
    rng = np.random.default_rng(6)
    for nprb, alloc, nports, nl, syms in ((273, slice(0, 273), 1, 1, [2]), (106, slice(10, 60), 2, 2, [2, 11]),
                                          (52, [0, 1, 2, 10, 11, 30], 1, 1, [3]), (25, slice(0, 25), 4, 4, [2, 3, 10, 11])):
        rb = np.zeros(nprb, np.uint8)
        rb[alloc] = 1
        sm = np.zeros(14, np.uint8)
        sm[syms] = 1
        g = ((rng.standard_normal((nports, 14, nprb * 12)) + 1j * rng.standard_normal((nports, 14, nprb * 12))) * 0.7).astype(np.complex64)
        a = (1, 3, False, 77, 0, 1.0, sm, rb, 0, 14, nl, g)
        (co, so), (cr, sr) = O.o_dmrs_pusch_estimate(*a), O.r_dmrs_pusch_estimate(*a)
        mask = np.repeat(rb.astype(bool), 12)
        assert np.abs(co[..., mask] - cr[..., mask]).max() < 1e-5 * np.abs(cr[..., mask]).max()
        assert np.all(np.abs(so[..., :4] - sr[..., :4]) <= 1e-5 * np.abs(sr[..., :4]))
        assert np.array_equal(so[..., 4], sr[..., 4])


def test_polar_chains_and_pdcch():
    rng = np.random.default_rng(7)
    cases = [(A + 24, 108 * AL, 9, 0) for A in (12, 40, 70, 128) for AL in (1, 2, 4, 8, 16) if A + 24 < 108 * AL] + [(56, 864, 9, 0)]
    cases += [(K, E, 10, ib) for K, E in ((18, 60), (25, 300), (31, 64), (100, 200), (500, 1500), (1023, 2000), (64, 8192), (22, 500)) for ib in (0, 1)]
    for K, E, nMax, ibil in cases:
        msg = rng.integers(0, 2, K, dtype=np.uint8)
        oo, rr = O.o_polar_encode_chain(K, E, nMax, ibil, msg), O.r_polar_encode_chain(K, E, nMax, ibil, msg)
        assert all(np.array_equal(a, b) for a, b in zip(oo, rr)), (K, E)
        for kind in range(3):
            if kind == 0:
                llr = (1 - 2 * oo[0].astype(np.int16)).astype(np.int8)
            elif kind == 1:
                llr = noisy(oo[0], 0.8, rng)
            else:
                llr = rng.integers(-120, 121, E).astype(np.int8)
                llr[rng.random(E) < 0.05] = 127
            od, rd = O.o_polar_decode_chain(K, E, nMax, ibil, llr), O.r_polar_decode_chain(K, E, nMax, ibil, llr)
            assert all(np.array_equal(a, b) for a, b in zip(od, rd)), (K, E, kind)
    for A, AL in ((12, 1), (40, 2), (70, 8), (128, 16)):
        pay = rng.integers(0, 2, A, dtype=np.uint8)
        assert np.array_equal(O.o_pdcch_encode(pay, 0x1234, 108 * AL), O.r_pdcch_encode(pay, 0x1234, 108 * AL))


def test_pbch_encoder():
    rng = np.random.default_rng(8)
    for i in range(60):
        L_max = [4, 8, 64][i % 3]
        a = (int(rng.integers(0, 1008)), int(rng.integers(0, L_max)), L_max, int(rng.integers(0, 2)), int(rng.integers(0, 1024)),
             int(rng.integers(0, 12 if L_max == 64 else 24)), rng.integers(0, 2, 32, dtype=np.uint8))
        assert np.array_equal(O.o_pbch_encode(*a), O.r_pbch_encode(*a)), a[:6]


def test_pusch_demodulator():
    """Oracle restatement vs the reference demodulation mapper (all modulations, abnormal noise variances) and vs the whole
    pusch_demodulator_impl (1-4 ports, both CDM settings, partial allocations). Tolerance: one quantisation step."""
    rng = np.random.default_rng(77)
    for mod in (1, 2, 4, 6, 8):
        n = 20000 + 3
        bits = rng.integers(0, 2, n * mod, dtype=np.uint8)
        assert np.abs(O.nr_modulate(bits, mod) - O.r_modulate(mod, bits)).max() < 3e-7  # the test-side mapper is the reference's
        x = O.nr_modulate(bits, mod) + ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.15).astype(np.complex64)
        nv = rng.uniform(0.005, 0.2, n).astype(np.float32)
        nv[::97], nv[5::101], nv[7::103], nv[11::107] = 0, np.inf, np.nan, -1
        o, r = O.o_demodulate_soft(mod, x, nv), O.r_demodulate_soft(mod, x, nv)
        diff = np.abs(o.astype(int) - r.astype(int))
        assert diff.max() <= 1 and (diff == 0).mean() > 0.9999, (mod, diff.max(), (diff == 0).mean())
    for mod, ports, cdm, type2 in ((8, 1, 2, 0), (6, 2, 1, 0), (4, 4, 2, 0), (2, 1, 1, 0), (1, 2, 2, 0), (6, 1, 1, 1), (4, 2, 3, 1), (8, 3, 2, 1)):
        nprb = 25
        nsc = nprb * 12
        rb = np.zeros(nprb, np.uint8)
        rb[3:20] = 1
        rb[21] = 1
        dm = np.zeros(14, np.uint8)
        dm[[2, 11]] = 1
        grid = (rng.standard_normal((ports, 14, nsc)) + 1j * rng.standard_normal((ports, 14, nsc))).astype(np.complex64)
        ce = (rng.standard_normal((ports, 14, nsc)) + 1j * rng.standard_normal((ports, 14, nsc))).astype(np.complex64)
        ce[0, 5, 40] = 0
        args = (0x4601, 935, mod, 1, 13, dm, type2, cdm, rb, grid, ce, 0.05)
        (o, eq, nv), r = O.o_pusch_demodulate(*args), O.r_pusch_demodulate(*args)
        diff = np.abs(o.astype(int) - r.astype(int))
        assert diff.max() <= 1 and (diff == 0).mean() > 0.99, (mod, ports, diff.max(), (diff == 0).mean())


def test_pusch_demodulator_placeholders_and_evm():
    """UCI on PUSCH in the demodulator (pusch_demodulator_impl.cpp:89-152): repetition placeholders in the descrambler and the EVM of
    the demodulation status, oracle vs reference. LLRs within one quantisation step (the reference's approximate reciprocal), EVM
    within 1e-3 relative (a hard decision that flips with such a step moves one symbol)."""
    rng = np.random.default_rng(78)
    for mod, ports, cdm in ((2, 1, 2), (4, 2, 2), (6, 1, 1), (8, 2, 2), (1, 1, 2)):
        nprb = 20
        nsc = nprb * 12
        rb = np.zeros(nprb, np.uint8)
        rb[2:19] = 1
        dm = np.zeros(14, np.uint8)
        dm[[2, 9]] = 1
        n_re = O.pusch_nof_re(0, 14, dm, 0, cdm, rb)
        bits = rng.integers(0, 2, n_re * mod, dtype=np.uint8)
        tx = O.nr_modulate(bits, mod)
        h = (rng.standard_normal((ports, 1, nsc)) + 1j * rng.standard_normal((ports, 1, nsc))).astype(np.complex64) * 0.8
        ce = np.ascontiguousarray(np.broadcast_to(h, (ports, 14, nsc)))
        grid = np.zeros((ports, 14, nsc), np.complex64)
        k = 0
        dmask = [(q % 2) < cdm for q in range(12)]
        for sy in range(14):
            for r in np.nonzero(rb)[0]:
                for q in range(12):
                    if dm[sy] and dmask[q]:
                        continue
                    grid[:, sy, r * 12 + q] = ce[:, sy, r * 12 + q] * tx[k]
                    k += 1
        assert k == n_re
        grid += ((rng.standard_normal(grid.shape) + 1j * rng.standard_normal(grid.shape)) * 0.05).astype(np.complex64)
        ph = np.sort(rng.choice(n_re, 37, replace=False)).astype(np.uint16) if mod >= 2 else np.zeros(0, np.uint16)
        args = (0x4601, 935, mod, 0, 14, dm, 0, cdm, rb, grid, ce, 0.01)
        (o, oe), (r, re_) = O.o_pusch_demodulate_ex(*args, placeholders=ph), O.r_pusch_demodulate_ex(*args, placeholders=ph)
        diff = np.abs(o.astype(int) - r.astype(int))
        assert diff.max() <= 1 and (diff == 0).mean() > 0.97, (mod, diff.max(), (diff == 0).mean())  # 97-99.6 % identical: approximate reciprocal
        assert re_ > 0 and abs(oe - re_) <= 1e-3 * re_, (mod, oe, re_)
        if ph.size:  # the placeholders matter: without them the LLRs of those elements differ
            o2, _ = O.o_pusch_demodulate_ex(*args, placeholders=(), want_evm=False)
            assert (o2 != o).sum() > 0 and set(np.nonzero(o2 != o)[0] // mod) <= set(int(x) for x in ph)


def test_channel_equalizer_standalone():
    """channel_equalizer_zf_impl through its factory vs the restatement. One layer: the scalar tail of the reference (the last
    nof_re % 8 elements, equalize_zf_1xn.h:120-158) agrees to a few ulp (its build contracts a*b+c), the AVX2 body within the error
    of _mm256_rcp_ps; two layers on two ports: scalar on both sides, differences only from contraction, scaled by the cancellation
    of the determinant. Dead estimates and a zero / negative noise variance give (0, +inf) on both sides."""
    rng = np.random.default_rng(4242)
    for npt, nl, nre in ((1, 1, 13), (2, 1, 260), (3, 1, 47), (4, 1, 1000), (2, 2, 9), (2, 2, 777)):
        for nvar_override in (None, 0.0, -1.0):
            y, h, nvar, _ = O.equalizer_case(rng, nre, npt, nl, snr_db=float(rng.uniform(0, 30)), dead=(3,))
            nvar = nvar if nvar_override is None else nvar_override
            txs = float(rng.choice([1.0, 0.5, 1.4142]))
            (z, nv), (zr, nvr) = O.o_channel_equalize(y, h, nvar, txs), O.r_channel_equalize(y, h, nvar, txs)
            assert np.array_equal(np.isinf(nv), np.isinf(nvr))
            if nvar_override is not None:
                assert np.all(np.isinf(nv)) and np.all(z == 0) and np.all(zr == 0)
                continue
            assert np.isinf(nv[:, 3]).all() and np.all(z[:, 3] == 0) and np.all(zr[:, 3] == 0)
            fin = ~np.isinf(nv)
            if nl == 1:
                tail = np.zeros(nre, bool)
                tail[nre // 8 * 8:] = True
                t = fin[0] & tail
                assert np.all(np.abs(z[0][t] - zr[0][t]) <= 1e-6 * (np.abs(zr[0][t]) + 1.0))
                assert np.all(np.abs(nv[0][t] - nvr[0][t]) <= 1e-6 * nvr[0][t])
                assert np.all(np.abs(z[fin] - zr[fin]) <= 4e-4 * np.abs(zr[fin]) + 1e-6)
                assert np.all(np.abs(nv[fin] - nvr[fin]) <= 4e-4 * nvr[fin])
            else:
                n0, n1 = (np.abs(h[0]) ** 2).sum(0), (np.abs(h[1]) ** 2).sum(0)
                cond = (n0 * n1) / np.maximum(n0 * n1 - np.abs((h[0].conj() * h[1]).sum(0)) ** 2, 1e-30)
                tol = 2e-6 * cond
                for l in range(2):
                    f = fin[l]
                    assert np.all(np.abs(z[l][f] - zr[l][f]) <= tol[f] * (np.abs(zr[l][f]) + 1.0))
                    assert np.all(np.abs(nv[l][f] - nvr[l][f]) <= tol[f] * nvr[l][f])


def test_pdsch_modulator_and_dmrs():
    """Modulation mapper, pdsch_modulator_impl (one layer, contiguous allocation: what 23.5 can do) and dmrs_pdsch_processor_impl
    against the oracle: bit-exact single-precision grids."""
    rng = np.random.default_rng(91)
    for mod in (1, 2, 4, 6, 8):
        bits = rng.integers(0, 2, 2000 * mod, dtype=np.uint8)
        assert np.array_equal(O.o_modulate(mod, bits).view(np.uint32), O.r_modulate(mod, bits).view(np.uint32))
    for (mod, nprb_grid, bwp_start, bwp_size, v0, v1, start, nof, dsyms, type2, cdm, nres, scaling, port, ngp) in [
            (8, 52, 0, 52, 0, 52, 0, 14, (2,), 0, 2, 0, 1.0, 0, 1), (6, 106, 10, 60, 5, 47, 2, 12, (2, 11), 0, 1, 2, 0.7, 1, 2),
            (4, 60, 4, 50, 0, 50, 1, 13, (3,), 1, 2, 1, 1.0, 3, 4), (2, 32, 0, 32, 7, 8, 0, 14, (2, 7), 1, 1, 4, float("nan"), 2, 3),
            (2, 275, 0, 275, 0, 275, 0, 14, (2, 3), 0, 2, 3, 0.5, 0, 1), (1, 40, 3, 30, 2, 20, 0, 14, (2,), 0, 2, 0, 1.0, 0, 1)]:
        dm = np.zeros(14, np.uint8)
        dm[list(dsyms)] = 1
        vrb = np.zeros(bwp_size, np.uint8)
        vrb[v0:v1] = 1
        reserved = [((rng.uniform(size=nprb_grid) < 0.5).astype(np.uint8), int(rng.integers(1, 4096)), int(rng.integers(1, 1 << 14))) for _ in range(nres)]
        pl = O.r_prb_indices(bwp_start, bwp_size, vrb, 0)
        nre = O.pdsch_nof_re(pl, start, nof, dm, type2, cdm, bwp_start, bwp_size, reserved)
        cw = rng.integers(0, 2, nre * mod, dtype=np.uint8)
        ref, pl2 = O.r_pdsch_modulate(0x1234, 77, scaling, 1, [mod], [cw], start, nof, dm, type2, cdm, bwp_start, bwp_size, vrb, 0, reserved, [port],
                                      nprb_grid, ngp)
        g = np.zeros_like(ref)
        assert O.o_pdsch_modulate(0x1234, 77, scaling, 1, [mod], [cw], start, nof, dm, type2, cdm, bwp_start, bwp_size, pl, reserved, [port], nprb_grid, g) == nre
        assert np.array_equal(pl, pl2) and np.array_equal(g.view(np.uint32), ref.view(np.uint32)), mod
    for (type2, nports, ref_pt, syms) in ((0, 4, 0, (2, 3)), (1, 6, 5, (2,)), (0, 8, 2, (2, 3, 10, 11)), (1, 12, 0, (4, 5))):
        nprb = 40
        rb = np.zeros(nprb, np.uint8)
        rb[ref_pt + 2: ref_pt + 20] = 1
        rb[30:33] = 1
        sm = np.zeros(14, np.uint8)
        sm[list(syms)] = 1
        ref = O.r_dmrs_pdsch_map(1, 13, ref_pt, type2, 321, 1, 1.4125, sm, rb, list(range(nports)), nports)
        g = np.zeros_like(ref)
        O.o_dmrs_pdsch_map(13, ref_pt, type2, 321, 1, 1.4125, sm, rb, list(range(nports)), g)
        assert np.array_equal(g.view(np.uint32), ref.view(np.uint32))


def test_ofh_iq_compression():
    """Both formats, every data width, odd and even PRB counts (the SIMD / scalar split of the quantiser), ties, clipping and
    overflow inputs, against the reference's generic, avx2 and avx512 classes."""
    rng = np.random.default_rng(77)
    for comp in (O.OFH_BFP, O.OFH_NONE):
        for w in range(1, 17):
            for nprb in (1, 2, 3, 17, 273):
                pl = rng.integers(0, 256, O.ofh_payload_bytes(nprb, w, comp), dtype=np.uint8)
                if comp == O.OFH_BFP:
                    pl[::1 + 3 * w] = rng.integers(0, 16 - w + 1, nprb)
                for impl, simd in (("avx2", True), ("avx512", True), ("generic", False)):
                    assert np.array_equal(O.o_ofh_iq_decompress(pl, nprb, w, simd, comp).view(np.uint32),
                                          O.r_ofh_iq_decompress(pl, nprb, w, impl, comp).view(np.uint32)), (comp, w, nprb, impl)
        for t in range(400):
            w, nprb = int(rng.integers(8, 17)), int(rng.integers(1, 8))
            gain = 32767 if comp == O.OFH_BFP else (1 << (w - 1)) - 1
            x = ((rng.standard_normal(nprb * 12) + 1j * rng.standard_normal(nprb * 12)) * 10 ** rng.uniform(-4, 0.5)).astype(np.complex64)
            if t % 7 == 0:
                x = (np.round(x.view(np.float32) * gain * 2) / 2 / gain).astype(np.float32).view(np.complex64)
            if t % 50 == 0:
                x.view(np.float32)[::5] = 1e12
            sc = float(rng.choice([1.0, 0.5, 0.37]))
            a = O.o_ofh_iq_compress(x, nprb, w, sc, comp)
            for impl in ("generic", "avx2", "avx512"):
                assert np.array_equal(a, O.r_ofh_iq_compress(x, nprb, w, sc, impl, comp)), (comp, t, w, nprb, sc, impl)
            # decompress(compress(x)) is x to within the quantisation step of the block
            if np.max(np.abs(x.view(np.float32))) * sc < 1.0:
                y = O.o_ofh_iq_decompress(a, nprb, w, True, comp)
                step = (2.0 ** a[::1 + 3 * w].astype(np.float64).repeat(24) / 32767) if comp == O.OFH_BFP else np.full(nprb * 24, 1.0 / gain)
                assert np.all(np.abs(y.view(np.float32) - x.view(np.float32) * np.float32(sc)) <= step * 1.001 + 1e-7), (comp, t, w)


def test_pdcch_processor():
    rng = np.random.default_rng(91)
    for (mapping, bs, bz, start, dur, fr, rbz, il, shift, cce, AL) in O.pdcch_cases(rng, 60):
        A = int(rng.integers(12, min(129, 108 * AL - 24)))
        pay = rng.integers(0, 2, A, dtype=np.uint8)
        rnti, nd, ndm, nr, slot = int(rng.integers(1, 65536)), int(rng.integers(0, 65536)), int(rng.integers(0, 65536)), int(rng.integers(0, 65536)), int(rng.integers(0, 20))
        ddb, xdb = float(rng.choice([0.0, 3.0, -1.5])), float(rng.choice([0.0, -3.0, 2.0]))
        g, rb = O.r_pdcch_process(mapping, bs, bz, start, dur, fr, rbz, il, shift, 1, slot, rnti, ndm, nd, nr, cce, AL, ddb, xdb, pay, bs + bz)
        out = np.zeros_like(g)
        assert O.o_pdcch_process(slot, rnti, nd, nr, ndm, bs if mapping == 0 else 0, xdb, ddb, pay, AL, start, dur, rb, out) == 54 * AL
        assert np.array_equal(out.view(np.uint32), g.view(np.uint32)), (mapping, bs, bz, dur, AL)


def test_ssb_processor():
    rng = np.random.default_rng(92)
    n = 0
    for (mu, sfn, slot, N_id, beta, ssb_idx, L_max, scs, kssb, off, case) in O.ssb_cases(rng, 80):
        pay = rng.integers(0, 2, 32, dtype=np.uint8)
        rc, g, l0, k0 = O.r_ssb_process(mu, sfn, slot, N_id, beta, ssb_idx, L_max, scs, kssb, off, case, pay, 106)
        assert rc == 0
        out = np.zeros_like(g)
        assert O.o_ssb_process(N_id, ssb_idx, L_max, 1 if slot >= (5 << mu) else 0, sfn, kssb, pay, k0, l0, beta, 106, out) == 0
        assert np.array_equal(out.view(np.uint32), g.view(np.uint32)), (case, N_id, ssb_idx, L_max)
        n += 1
    assert n == 80


def test_nzp_csi_rs_generator():
    rng = np.random.default_rng(93)
    for (row, nports, k, cdm, dens, start_rb, nof_rb, l0, slot, scr, amp) in O.csi_rs_cases(rng, 96):
        l0 = min(l0, 12)
        g, bes, rm, sm = O.r_csi_rs_map(1, slot, start_rb, nof_rb, row, k, l0, 0, cdm, dens, scr, amp, nports, 80)
        out = np.zeros_like(g)
        assert O.o_csi_rs_map(slot, scr, amp, start_rb, nof_rb, bes, row, cdm, dens, list(range(nports)), rm, sm, 80, out) == 0
        assert np.array_equal(out.view(np.uint32), g.view(np.uint32)), (row, dens, start_rb, nof_rb)


def test_mixed_slot_table_is_the_reference_calculators():
    """bench_legs.MIXED_PDUS (the mixed-slot leg of bench.py): TBS and base graph of every PDU are what the reference's
    tbs_calculator_calculate (lib/scheduler/support/tbs_calculator.cpp) and get_ldpc_base_graph (ldpc_base_graph.h:38) return for its
    PRBs, modulation and code rate; the PDUs tile the 273-PRB grid. The headline allocation gives BASELINE's 319 784 bits."""
    import bench_legs as BL
    nxt = 0
    for rb0, nprb, mod, tbs, bg, R in BL.MIXED_PDUS:
        assert rb0 == nxt
        nxt += nprb
        assert O.r_tbs_calculate(14, 12, mod, R, 1, nprb) == tbs
        assert O.r_ldpc_base_graph(R, tbs) == bg
    assert nxt == 273
    assert O.r_tbs_calculate(14, 12, 8, 948, 1, 273) == 319784
