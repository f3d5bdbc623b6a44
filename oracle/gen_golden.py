#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: generates the golden fixtures under tests/golden/ by running THE REFERENCE ITSELF
(srsRAN_Project 23.5 compiled in place into oracle/_ref/libref_capi.so by oracle/build_ref.sh).

Each .npz holds seeded inputs and the outputs the reference produced for them (AVX2 implementations where the reference
has several).  The fixtures are data only; they let the GPU box -- which has neither /root/reference nor oracle/_ref
sources -- check the CPU oracle (tests/test_oracle_golden.py) and the HIP kernels against reference-produced vectors.
Run:  bash oracle/build_ref.sh && python oracle/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
rng = np.random.default_rng(20231005)


def noisy(cw, sigma):
    y = (1.0 - 2.0 * (cw & 1)) + sigma * rng.standard_normal(cw.size)
    return np.round(np.clip(4 * y, -20, 20) / 20 * 120).astype(np.int8)


ONLY = set(sys.argv[1:])  # optional: names of the fixtures to (re)write; the random stream is consumed identically either way


def save(name, **arrays):
    if ONLY and name not in ONLY:
        return
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %7.1f KiB" % (name, os.path.getsize(path) / 1024))


# ------------------------------------------------------------------ CRC
d = {}
for i, (poly, n) in enumerate([(p, n) for p in range(5) for n in (1, 24, 100, 1001, 8424)]):
    bits = rng.integers(0, 2, n, dtype=np.uint8)
    d["bits_%d" % i] = bits
    d["meta_%d" % i] = np.array([poly, O.r_crc_bits(poly, bits)], dtype=np.int64)
save("crc", **d)

# ------------------------------------------------------------------ LDPC encoder / decoder (subset of the 102 graphs)
d = {}
dec = O.RefLdpcDecoder("avx2")
i = 0
for bg in (1, 2):
    for Z in (2, 7, 24, 52, 104, 208, 352, 384):
        K, NS = O.BG_K[bg] * Z, O.BG_NS[bg] * Z
        nf = int(rng.integers(0, max(1, Z // 2)))
        poly, nb = (O.CRC24B, 24) if K - nf > 60 else (O.CRC16, 16)
        if K - nf <= nb + 2:
            nf = 0
        msg = rng.integers(0, 2, K, dtype=np.uint8)
        c = O.r_crc_bits(poly, msg[:K - nf - nb])
        msg[K - nf - nb:K - nf] = [(c >> (nb - 1 - j)) & 1 for j in range(nb)]
        if nf:
            msg[K - nf:] = 254
        cw = O.r_ldpc_encode(bg, Z, msg, NS, "avx2")
        L = [NS, K + 2 * Z, (K + 2 * Z + NS) // 2 // Z * Z][i % 3]
        llr = noisy(cw[:L], [0.55, 0.3, 0.8][i % 3])
        if nf:
            llr[K - 2 * Z - nf:K - 2 * Z] = 127
        rows = []
        for crc, mi in ((poly, 6), (-1, 2), (poly, 1)):
            it, bits = dec.decode(bg, Z, llr, nf, crc, mi)
            rows.append(np.concatenate([[crc, mi, it], bits]).astype(np.int64))
        d["msg_%d" % i], d["cw_%d" % i], d["llr_%d" % i] = msg, cw, llr
        d["meta_%d" % i] = np.array([bg, Z, nf, L], dtype=np.int64)
        d["dec_%d" % i] = np.stack(rows)
        i += 1
save("ldpc_enc_dec", **d)

# ------------------------------------------------------------------ rate matcher / dematcher
d = {}
i = 0
for bg, Z in ((1, 384), (2, 208), (1, 16), (2, 7)):
    N, K = O.BG_NS[bg] * Z, O.BG_K[bg] * Z
    for rv in range(4):
        for mod, Nref in ((1, 0), (2, N - 3 * Z), (4, 0), (6, (2 * N) // 3), (8, 0)):
            if Z > 100 and (rv, mod) not in ((0, 8), (2, 2), (3, 6)):
                continue  # keep the fixture small: three large-Z cases per graph
            nf = int(rng.integers(0, Z))
            cb = rng.integers(0, 2, N, dtype=np.uint8)
            if nf:
                cb[K - 2 * Z - nf:K - 2 * Z] = 254
            E = mod * int(rng.integers(K // (2 * mod), (2 * N) // mod))
            rm = O.r_rate_match(bg, Z, rv, mod, Nref, nf, cb, E)
            llr = rng.integers(-120, 121, E).astype(np.int8)
            sb = rng.integers(-120, 121, N).astype(np.int8)
            d["cb_%d" % i], d["rm_%d" % i], d["llr_%d" % i], d["sb_%d" % i] = cb, rm, llr, sb
            d["rdm_new_%d" % i] = O.r_rate_dematch(bg, Z, rv, mod, Nref, nf, 1, llr, sb, "avx2")
            d["rdm_comb_%d" % i] = O.r_rate_dematch(bg, Z, rv, mod, Nref, nf, 0, llr, sb, "avx2")
            d["meta_%d" % i] = np.array([bg, Z, rv, mod, Nref, nf, E], dtype=np.int64)
            i += 1
save("ldpc_rate_match", **d)

# ------------------------------------------------------------------ PDSCH encoder / PUSCH decoder with HARQ
d = {}
for i, (bg, mod, nl, nprb, tbs, sigma) in enumerate([(2, 2, 1, 20, 1032, 1.1), (1, 4, 1, 40, 15880, 0.62), (1, 6, 2, 25, 40976, 0.45),
                                                     (2, 2, 1, 2, 24, 0.9)]):
    nsym = nprb * 156 * nl
    tb = rng.integers(0, 256, tbs // 8, dtype=np.uint8)
    rvs = [0, 2, 3, 1]
    cws = np.stack([O.r_pdsch_encode(bg, rv, mod, 0, nl, nsym, tb, "avx2") for rv in rvs])
    llrs = np.stack([noisy(c, sigma) for c in cws])
    pd = O.RefPuschDecoder("avx2")
    ok, tbo, mm = pd.decode_sequence(bg, mod, 0, nl, nsym, tbs // 8, rvs, llrs, 6, True)
    d["tb_%d" % i], d["cw_%d" % i], d["llr_%d" % i] = tb, cws, llrs
    d["meta_%d" % i] = np.array([bg, mod, nl, nsym, tbs], dtype=np.int64)
    d["res_%d" % i] = np.array([[int(o), a, b] for o, (a, b) in zip(ok, mm)], dtype=np.int64)
    d["tbo_%d" % i] = tbo
save("sch_chain", **d)

# ------------------------------------------------------------------ DFT / OFDM
d = {}
for i, N in enumerate((128, 384, 512, 1536)):
    x = (rng.uniform(-1, 1, N) + 1j * rng.uniform(-1, 1, N)).astype(np.complex64)
    d["x_%d" % i], d["fwd_%d" % i], d["inv_%d" % i] = x, O.r_dft(x, False), O.r_dft(x, True)
save("dft", **d)
d = {}
for i, (mu, rb, N, wo, fc, slot) in enumerate([(0, 25, 512, 18, 2.6e9, 0), (1, 51, 1024, 36, 3.5e9, 1)]):
    cfg = O.OfdmCfg(mu, rb, N, wo, 0.5, fc)
    ns = O.o_ofdm_slot_size(cfg, slot)
    x = ((rng.standard_normal(ns) + 1j * rng.standard_normal(ns)) * 0.7).astype(np.complex64)
    g = (rng.standard_normal((14, rb * 12)) + 1j * rng.standard_normal((14, rb * 12))).astype(np.complex64)
    d["x_%d" % i], d["grid_%d" % i] = x, O.r_ofdm_demod_slot(cfg, slot, x)
    d["g_%d" % i], d["y_%d" % i] = g, O.r_ofdm_mod_slot(O.OfdmCfg(mu, rb, N, 0, 0.01, fc), slot, g, ns)
    d["meta_%d" % i] = np.array([mu, rb, N, wo, fc, slot], dtype=np.float64)
save("ofdm", **d)

# ------------------------------------------------------------------ DM-RS PUSCH estimator
d = {}
for i, (nprb, alloc, nports, nl, syms, mu, slot, scr, nscid, scaling) in enumerate([
        (25, slice(0, 25), 2, 1, [2], 1, 3, 77, 0, 1.0), (52, slice(10, 40), 1, 2, [2, 11], 1, 17, 1000, 1, 0.7071),
        (30, [0, 1, 2, 10, 11, 20], 2, 1, [2, 7, 11], 0, 9, 5, 0, 1.0)]):
    rb = np.zeros(nprb, np.uint8)
    rb[alloc] = 1
    sm = np.zeros(14, np.uint8)
    sm[syms] = 1
    g = ((rng.standard_normal((nports, 14, nprb * 12)) + 1j * rng.standard_normal((nports, 14, nprb * 12))) * 0.7).astype(np.complex64)
    ce, sc = O.r_dmrs_pusch_estimate(mu, slot, False, scr, nscid, scaling, sm, rb, 0, 14, nl, g)
    mask = np.repeat(rb.astype(bool), 12)
    d["grid_%d" % i], d["ce_%d" % i], d["sc_%d" % i] = g, ce[..., mask], sc
    d["rb_%d" % i], d["sm_%d" % i] = rb, sm
    d["meta_%d" % i] = np.array([mu, slot, scr, nscid, scaling, nl], dtype=np.float64)
save("dmrs_pusch_estimator", **d)

# ------------------------------------------------------------------ polar chains / PDCCH
d = {}
for i, (K, E, nMax, ibil) in enumerate([(36, 108, 9, 0), (64, 216, 9, 0), (94, 432, 9, 0), (152, 1728, 9, 0), (56, 864, 9, 0),
                                        (20, 100, 10, 1), (25, 300, 10, 0), (100, 200, 10, 1), (500, 1500, 10, 0), (300, 400, 10, 1)]):
    msg = rng.integers(0, 2, K, dtype=np.uint8)
    rm, al, en = O.r_polar_encode_chain(K, E, nMax, ibil, msg)
    llr = noisy(rm, 0.8)
    m2, dem, u = O.r_polar_decode_chain(K, E, nMax, ibil, llr)
    d["msg_%d" % i], d["rm_%d" % i], d["alloc_%d" % i], d["enc_%d" % i] = msg, rm, al, en
    d["llr_%d" % i], d["dec_msg_%d" % i], d["dem_%d" % i], d["u_%d" % i] = llr, m2, dem, u
    d["meta_%d" % i] = np.array([K, E, nMax, ibil], dtype=np.int64)
for i, (A, AL) in enumerate([(12, 1), (40, 2), (70, 8), (128, 16)]):
    pay = rng.integers(0, 2, A, dtype=np.uint8)
    rnti = int(rng.integers(0, 65536))
    d["pdcch_pay_%d" % i], d["pdcch_out_%d" % i] = pay, O.r_pdcch_encode(pay, rnti, 108 * AL)
    d["pdcch_meta_%d" % i] = np.array([A, 108 * AL, rnti], dtype=np.int64)
for i in range(6):
    L_max = [4, 8, 64][i % 3]
    a = (int(rng.integers(0, 1008)), int(rng.integers(0, L_max)), L_max, int(rng.integers(0, 2)), int(rng.integers(0, 1024)),
         int(rng.integers(0, 12 if L_max == 64 else 24)))
    pay = rng.integers(0, 2, 32, dtype=np.uint8)
    d["pbch_meta_%d" % i], d["pbch_pay_%d" % i], d["pbch_out_%d" % i] = np.array(a, dtype=np.int64), pay, O.r_pbch_encode(*a, pay)
save("polar", **d)

# ------------------------------------------------------------------ PUSCH demodulator: soft demapper alone + whole block
d = {}
for i, mod in enumerate((1, 2, 4, 6, 8)):
    n = 1200 + 3  # not a multiple of the AVX2 batch: the reference's scalar tail runs too
    bits = rng.integers(0, 2, n * mod, dtype=np.uint8)
    x = O.nr_modulate(bits, mod) + ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.12).astype(np.complex64)
    nv = rng.uniform(0.004, 0.3, n).astype(np.float32)
    nv[::97], nv[5::101], nv[11::107] = 0, np.inf, -1  # the reference maps all of these to LLR 0
    d["dm_sym_%d" % i], d["dm_nv_%d" % i], d["dm_llr_%d" % i] = x, nv, O.r_demodulate_soft(mod, x, nv)
    d["dm_meta_%d" % i] = np.array([mod], dtype=np.int64)
for i, (mod, ports, cdm, nprb_grid, start, nof, dsyms) in enumerate([(8, 1, 2, 12, 0, 14, (2,)), (6, 2, 1, 10, 1, 13, (2, 11)), (4, 4, 2, 8, 2, 10, (3, 7)),
                                                                      (2, 1, 1, 11, 0, 14, (2, 7, 11)), (1, 2, 2, 6, 0, 12, (2,)),
                                                                      (8, 2, 2, 9, 0, 14, (2, 11))]):
    nsc = nprb_grid * 12
    rb = (rng.uniform(size=nprb_grid) < 0.8).astype(np.uint8)
    rb[0] = 1
    dm = np.zeros(14, np.uint8)
    dm[list(dsyms)] = 1
    n_re = O.pusch_nof_re(start, nof, dm, 0, cdm, rb)
    bits = rng.integers(0, 2, n_re * mod, dtype=np.uint8)
    h = ((rng.standard_normal((ports, 1, nsc)) + 1j * rng.standard_normal((ports, 1, nsc))) * 0.7).astype(np.complex64) * np.ones((1, 14, 1), np.complex64)
    grid = ((rng.standard_normal((ports, 14, nsc)) + 1j * rng.standard_normal((ports, 14, nsc))) * 0.05).astype(np.complex64)
    x = O.nr_modulate(bits, mod)
    k = 0
    for sy in range(start, start + nof):  # place the symbols where the demodulator will look for them
        for r in range(nprb_grid):
            if not rb[r]:
                continue
            for q in range(12):
                if dm[sy] and (q % 2) < cdm:
                    continue
                grid[:, sy, r * 12 + q] += h[:, sy, r * 12 + q] * x[k]
                k += 1
    assert k == n_re
    ce = (h + ((rng.standard_normal(h.shape) + 1j * rng.standard_normal(h.shape)) * 0.01)).astype(np.complex64)
    ce[0, start, 5] = 0  # a dead estimate: with one port the reference outputs LLR 0 there
    rnti, n_id, noise_var = int(rng.integers(1, 65536)), int(rng.integers(0, 1024)), 0.005
    d["grid_%d" % i], d["ce_%d" % i], d["rb_%d" % i], d["dm_%d" % i] = grid, ce, rb, dm
    d["meta_%d" % i] = np.array([rnti, n_id, mod, start, nof, cdm, noise_var], dtype=np.float64)
    d["llr_%d" % i] = O.r_pusch_demodulate(rnti, n_id, mod, start, nof, dm, 0, cdm, rb, grid, ce, noise_var)
    d["bits_%d" % i] = bits
save("pusch_demod", **d)

# ------------------------------------------------------------------ PDSCH modulator + PDSCH DM-RS
d = {}
for i, mod in enumerate((1, 2, 4, 6, 8)):
    bits = rng.integers(0, 2, 509 * mod, dtype=np.uint8)
    d["map_bits_%d" % i], d["map_sym_%d" % i], d["map_meta_%d" % i] = bits, O.r_modulate(mod, bits), np.array([mod], dtype=np.int64)
for i, (mod, nprb_grid, bwp_start, bwp_size, v0, v1, start, nof, dsyms, type2, cdm, nres, scaling, port, ngp) in enumerate([
        (8, 24, 0, 24, 0, 24, 0, 14, (2,), 0, 2, 0, 1.0, 0, 1), (6, 40, 6, 30, 3, 25, 2, 12, (2, 11), 0, 1, 2, 0.7, 1, 2),
        (4, 30, 4, 26, 0, 26, 1, 13, (3,), 1, 2, 1, 1.0, 3, 4), (2, 20, 0, 20, 7, 8, 0, 14, (2, 7), 1, 1, 4, float("nan"), 2, 3),
        (1, 16, 2, 12, 1, 9, 0, 14, (2,), 0, 2, 0, 1.0, 0, 1)]):
    dm = np.zeros(14, np.uint8)
    dm[list(dsyms)] = 1
    vrb = np.zeros(bwp_size, np.uint8)
    vrb[v0:v1] = 1
    reserved = [((rng.uniform(size=nprb_grid) < 0.5).astype(np.uint8), int(rng.integers(1, 4096)), int(rng.integers(1, 1 << 14))) for _ in range(nres)]
    pl = O.r_prb_indices(bwp_start, bwp_size, vrb, 0)
    nre = O.pdsch_nof_re(pl, start, nof, dm, type2, cdm, bwp_start, bwp_size, reserved)
    cw = rng.integers(0, 2, nre * mod, dtype=np.uint8)
    rnti, n_id = int(rng.integers(1, 65536)), int(rng.integers(0, 1024))
    grid, pl2 = O.r_pdsch_modulate(rnti, n_id, scaling, 1, [mod], [cw], start, nof, dm, type2, cdm, bwp_start, bwp_size, vrb, 0, reserved, [port],
                                   nprb_grid, ngp)
    d["pm_cw_%d" % i], d["pm_grid_%d" % i], d["pm_prb_%d" % i], d["pm_dm_%d" % i] = cw, grid, pl2, dm
    d["pm_res_prb_%d" % i] = np.array([r[0] for r in reserved], dtype=np.uint8).reshape(nres, nprb_grid)
    d["pm_res_re_%d" % i] = np.array([[r[1], r[2]] for r in reserved], dtype=np.int64).reshape(nres, 2)
    d["pm_meta_%d" % i] = np.array([rnti, n_id, scaling, mod, start, nof, type2, cdm, bwp_start, bwp_size, port, nprb_grid, ngp], dtype=np.float64)
for i, (type2, nports, ref_pt, syms, nprb) in enumerate([(0, 4, 0, (2, 3), 24), (1, 6, 5, (2,), 30), (0, 8, 2, (2, 3, 10, 11), 20), (1, 12, 0, (4, 5), 12)]):
    rb = np.zeros(nprb, np.uint8)
    rb[ref_pt + 1: ref_pt + 9] = 1
    rb[nprb - 2] = 1
    sm = np.zeros(14, np.uint8)
    sm[list(syms)] = 1
    slot, scr, nscid, amp = int(rng.integers(0, 20)), int(rng.integers(0, 65536)), int(rng.integers(0, 2)), float(rng.uniform(0.5, 2.0))
    d["dd_grid_%d" % i] = O.r_dmrs_pdsch_map(1, slot, ref_pt, type2, scr, nscid, amp, sm, rb, list(range(nports)), nports)
    d["dd_rb_%d" % i], d["dd_sm_%d" % i] = rb, sm
    d["dd_meta_%d" % i] = np.array([slot, ref_pt, type2, scr, nscid, amp, nports], dtype=np.float64)
save("pdsch_mod", **d)

# ---------------------------------------------------------------------- rx_softbuffer_pool reservation traces
d = {"n": np.array(4)}
for i, (ms, mc, ex) in enumerate([(4, 20, 8), (2, 9, 3), (6, 14, 30), (3, 100, 0)]):
    ops = O.pool_trace(500 + i, 3000)
    d["cfg_%d" % i], d["ops_%d" % i], d["res_%d" % i] = np.array([ms, mc, ex]), ops.astype(np.int32), O.r_pool_run(ops, ms, mc, ex).astype(np.int32)
save("harq_pool", **d)

# ---------------------------------------------------------------------- Open Fronthaul IQ (de)compression: BFP and uncompressed
d = {}
cases = [(1, 9, 273), (1, 9, 51), (1, 14, 106), (1, 12, 7), (1, 16, 25), (1, 8, 2), (1, 4, 5), (1, 1, 3), (0, 16, 273), (0, 9, 106), (0, 12, 3),
         (0, 8, 1), (0, 5, 4)]
d["cases"] = np.array(cases)
for i, (comp, w, nprb) in enumerate(cases):
    pl = rng.integers(0, 256, O.ofh_payload_bytes(nprb, w, comp), dtype=np.uint8)
    if comp == O.OFH_BFP:
        pl[::1 + 3 * w] = rng.integers(0, 16 - w + 1, nprb)
    d["dec_payload_%d" % i] = pl
    d["dec_simd_%d" % i] = O.r_ofh_iq_decompress(pl, nprb, w, "avx2", comp)
    d["dec_generic_%d" % i] = O.r_ofh_iq_decompress(pl, nprb, w, "generic", comp)
    if w >= 8:
        x = ((rng.standard_normal(nprb * 12) + 1j * rng.standard_normal(nprb * 12)) * 10 ** rng.uniform(-3, 0.2, nprb).repeat(12)).astype(np.complex64)
        xr = x.view(np.float32)
        g = 32767 if comp == O.OFH_BFP else (1 << (w - 1)) - 1
        xr[::11] = np.round(xr[::11] * g * 2) / 2 / g  # exact .5 ties after scaling
        d["cmp_in_%d" % i] = x
        for sc in (1.0, 0.37):
            d["cmp_out_%d_%d" % (i, int(sc * 100))] = O.r_ofh_iq_compress(x, nprb, w, sc, "avx2", comp)
            assert np.array_equal(d["cmp_out_%d_%d" % (i, int(sc * 100))], O.r_ofh_iq_compress(x, nprb, w, sc, "generic", comp))
save("ofh_iq", **d)

# ---------------------------------------------------------------------- PDCCH processor (reference grids + the PRB masks of its CCE mapping)
d = {}
cases = O.pdcch_cases(np.random.default_rng(2024), 24)
d["n"] = np.array(len(cases))
for i, (mapping, bs, bz, start, dur, fr, rbz, il, shift, cce, AL) in enumerate(cases):
    A = int(rng.integers(12, min(129, 108 * AL - 24)))
    pay = rng.integers(0, 2, A, dtype=np.uint8)
    rnti, nd, ndm, nr, slot = int(rng.integers(1, 65536)), int(rng.integers(0, 65536)), int(rng.integers(0, 65536)), int(rng.integers(0, 65536)), int(rng.integers(0, 20))
    ddb, xdb = float(rng.choice([0.0, 3.0, -1.5])), float(rng.choice([0.0, -3.0, 2.0]))
    g, rb = O.r_pdcch_process(mapping, bs, bz, start, dur, fr, rbz, il, shift, 1, slot, rnti, ndm, nd, nr, cce, AL, ddb, xdb, pay, bs + bz)
    d["grid_%d" % i], d["rb_%d" % i], d["pay_%d" % i] = g, rb, pay
    d["meta_%d" % i] = np.array([slot, rnti, nd, nr, ndm, bs if mapping == 0 else 0, xdb, ddb, AL, start, dur, mapping], dtype=np.float64)
save("pdcch_proc", **d)

# ---------------------------------------------------------------------- SS/PBCH block processor (reference grids + the positions it derived)
d = {}
cases = O.ssb_cases(np.random.default_rng(77), 16)
k = 0
for (mu, sfn, slot, N_id, beta, ssb_idx, L_max, scs, kssb, off, case) in cases:
    pay = rng.integers(0, 2, 32, dtype=np.uint8)
    rc, g, l0, k0 = O.r_ssb_process(mu, sfn, slot, N_id, beta, ssb_idx, L_max, scs, kssb, off, case, pay, 106)
    assert rc == 0
    d["grid_%d" % k], d["pay_%d" % k] = g, pay
    d["meta_%d" % k] = np.array([N_id, ssb_idx, L_max, 1 if slot >= (5 << mu) else 0, sfn, kssb, k0, l0, beta, case], dtype=np.float64)
    k += 1
d["n"] = np.array(k)
save("ssb_proc", **d)

# ---------------------------------------------------------------------- NZP-CSI-RS generator (reference grids + the patterns of get_csi_rs_pattern)
d = {}
cases = O.csi_rs_cases(np.random.default_rng(31), 24)
for i, (row, nports, k, cdm, dens, start_rb, nof_rb, l0, slot, scr, amp) in enumerate(cases):
    l0 = min(l0, 12)
    g, bes, rm, sm = O.r_csi_rs_map(1, slot, start_rb, nof_rb, row, k, l0, 0, cdm, dens, scr, amp, nports, 80)
    d["grid_%d" % i], d["rm_%d" % i], d["sm_%d" % i] = g, rm[:nports], sm[:nports]
    d["meta_%d" % i] = np.array([slot, scr, amp, start_rb, nof_rb, bes[0], bes[1], bes[2], row, cdm, dens, nports], dtype=np.float64)
d["n"] = np.array(len(cases))
save("csi_rs", **d)

# ---------------------------------------------------------------------- zero-forcing equalizer on its own (reference outputs, AVX2 1 x N / scalar 2 x 2)
d = {}
erng = np.random.default_rng(5150)
EQ_CASES = [(1, 1, 300, 1.0), (2, 1, 301, 0.5), (3, 1, 77, 2.0), (4, 1, 1203, 1.0), (2, 2, 300, 1.0), (2, 2, 1201, 0.7071)]
for i, (npt, nl, nre, txs) in enumerate(EQ_CASES):
    y, h, nvar, x = O.equalizer_case(erng, nre, npt, nl, dead=(5, nre - 1))
    z, nv = O.r_channel_equalize(y, h, nvar, txs)
    d["y_%d" % i], d["h_%d" % i], d["z_%d" % i], d["nv_%d" % i] = y, h, z, nv
    d["meta_%d" % i] = np.array([nvar, txs], dtype=np.float64)
d["n"] = np.array(len(EQ_CASES))
save("channel_equalizer", **d)
