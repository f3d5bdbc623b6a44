"""GPU parity for the polar chains (encode: allocate+encode+rate-match; decode: dematch+SSC+deallocate) and the PDCCH
encoder vs the CPU oracle -- bit-exact, including every intermediate tap."""
import numpy as np
import pytest

from oracle_lib import (o_pbch_encode, o_pdcch_encode, o_polar_decode_chain, o_polar_encode_chain, o_polar_scl_decode)
from oracle_lib import o_polar_interleave as O_interleave

pytestmark = pytest.mark.gpu


def polar_cases():
    cases = []
    for A in (12, 40, 70, 140):  # PDCCH: K = A + 24, E = 108 * AL (BASELINE config C4)
        for AL in (1, 2, 4, 8, 16):
            if A + 24 < 108 * AL:
                cases.append((A + 24, 108 * AL, 9, 0))
    cases.append((56, 864, 9, 0))  # PBCH
    for K, E in ((18, 60), (20, 100), (25, 300), (31, 64), (40, 100), (100, 200), (200, 1000), (500, 1500), (1023, 2000),
                 (64, 8192), (300, 400), (22, 500), (19, 29)):  # uplink (UCI) incl. parity-check bits and all RM modes
        for ibil in (0, 1):
            cases.append((K, E, 10, ibil))
    return cases


@pytest.mark.parametrize("K,E,nMax,ibil", polar_cases())
def test_polar_chains(ctx, K, E, nMax, ibil):
    import torch
    import miphy
    rng = np.random.default_rng(K * 7 + E)
    code = miphy.PolarCode(K, E, nMax, ibil)
    n_log, N, nPC = code.info()
    nb = 6
    msgs = rng.integers(0, 2, (nb, K), dtype=np.uint8)
    m_d = torch.from_numpy(msgs.reshape(-1)).cuda()
    out_d = torch.zeros(nb * E, dtype=torch.uint8, device="cuda")
    al_d = torch.zeros(nb * N, dtype=torch.uint8, device="cuda")
    en_d = torch.zeros(nb * N, dtype=torch.uint8, device="cuda")
    ctx.polar_encode_batch(code, nb, m_d, out_d, al_d, en_d)
    torch.cuda.synchronize()
    out, al, en = out_d.cpu().numpy().reshape(nb, E), al_d.cpu().numpy().reshape(nb, N), en_d.cpu().numpy().reshape(nb, N)
    exp = [o_polar_encode_chain(K, E, nMax, ibil, msgs[i]) for i in range(nb)]
    for i in range(nb):
        assert exp[i][1].size == N
        assert np.array_equal(al[i], exp[i][1]) and np.array_equal(en[i], exp[i][2]) and np.array_equal(out[i], exp[i][0]), (K, E, i)
    # decode: noiseless +-1 (polar_chain_test.cpp:192-210), AWGN, and arbitrary LLRs incl. infinities
    llrs = np.zeros((nb, E), dtype=np.int8)
    for i in range(nb):
        if i < 2:
            llrs[i] = 1 - 2 * exp[i][0].astype(np.int16)
        elif i < 4:
            y = (1.0 - 2.0 * exp[i][0]) + [0.7, 1.0][i - 2] * rng.standard_normal(E)
            llrs[i] = np.round(np.clip(4 * y, -20, 20) / 20 * 120)
        else:
            v = rng.integers(-120, 121, E)
            v[rng.random(E) < 0.1] = 0
            v[rng.random(E) < 0.05] = 127
            v[rng.random(E) < 0.05] = -127
            llrs[i] = v
    l_d = torch.from_numpy(llrs.reshape(-1)).cuda()
    msg_d = torch.zeros(nb * K, dtype=torch.uint8, device="cuda")
    dem_d = torch.zeros(nb * N, dtype=torch.int8, device="cuda")
    u_d = torch.zeros(nb * N, dtype=torch.uint8, device="cuda")
    ctx.polar_decode_batch(code, nb, l_d, msg_d, dem_d, u_d)
    torch.cuda.synchronize()
    got_m, got_d, got_u = msg_d.cpu().numpy().reshape(nb, K), dem_d.cpu().numpy().reshape(nb, N), u_d.cpu().numpy().reshape(nb, N)
    for i in range(nb):
        em, ed, eu = o_polar_decode_chain(K, E, nMax, ibil, llrs[i])
        assert np.array_equal(got_d[i], ed), (K, E, i, "dematch")
        assert np.array_equal(got_u[i], eu), (K, E, i, "decode")
        assert np.array_equal(got_m[i], em), (K, E, i, "deallocate")
        if i < 2:
            assert np.array_equal(got_m[i], msgs[i])


@pytest.mark.parametrize("K,E,nMax,ibil", [(64, 108, 9, 0), (64, 432, 9, 0), (164, 1728, 9, 0), (56, 864, 9, 0), (25, 300, 10, 1), (1023, 2000, 10, 0)])
@pytest.mark.parametrize("n", [4099, 8195])
def test_polar_decode_large_batches_share_wavefronts(ctx, K, E, nMax, ibil, n):
    """Batches that fill the chip decode two (n >= 4096) or four (n >= 8192) codewords per wavefront in lockstep on the shared schedule
    (polar_decode_kernel<2>, <4>); ragged last wavefronts. Every codeword -- message, dematcher tap, decoder tap -- must equal the oracle's
    (the reference's SSC decoder), whatever its neighbours in the wavefront are: noiseless, AWGN and arbitrary LLRs with infinities side by side."""
    import torch
    import miphy
    rng = np.random.default_rng(K * 11 + E + n)
    code = miphy.PolarCode(K, E, nMax, ibil)
    _, N, _ = code.info()
    pool = 23
    msgs = rng.integers(0, 2, (pool, K), dtype=np.uint8)
    llrs = np.zeros((pool, E), dtype=np.int8)
    for i in range(pool):
        cwd = o_polar_encode_chain(K, E, nMax, ibil, msgs[i])[0]
        if i % 3 == 0:
            llrs[i] = 1 - 2 * cwd.astype(np.int16)
        elif i % 3 == 1:
            y = (1.0 - 2.0 * cwd) + [0.6, 0.9, 1.3][i % 9 // 3] * rng.standard_normal(E)
            llrs[i] = np.round(np.clip(4 * y, -20, 20) / 20 * 120)
        else:
            v = rng.integers(-120, 121, E)
            v[rng.random(E) < 0.1] = 0
            v[rng.random(E) < 0.05] = 127
            v[rng.random(E) < 0.05] = -127
            llrs[i] = v
    exp = [o_polar_decode_chain(K, E, nMax, ibil, llrs[i]) for i in range(pool)]
    idx = rng.integers(0, pool, n)
    l_d = torch.from_numpy(llrs[idx].reshape(-1)).cuda()
    msg_d = torch.zeros(n * K, dtype=torch.uint8, device="cuda")
    dem_d = torch.zeros(n * N, dtype=torch.int8, device="cuda")
    u_d = torch.zeros(n * N, dtype=torch.uint8, device="cuda")
    ctx.polar_decode_batch(code, n, l_d, msg_d, dem_d, u_d)
    torch.cuda.synchronize()
    got_m, got_d, got_u = msg_d.cpu().numpy().reshape(n, K), dem_d.cpu().numpy().reshape(n, N), u_d.cpu().numpy().reshape(n, N)
    em, ed, eu = (np.stack([e[j] for e in exp]) for j in range(3))
    assert np.array_equal(got_d, ed[idx]), "dematch"
    assert np.array_equal(got_u, eu[idx]), "decode"
    assert np.array_equal(got_m, em[idx]), "deallocate"


def test_pdcch_encoder_batch(ctx):
    import torch
    rng = np.random.default_rng(41)
    for A, AL in ((12, 1), (40, 2), (39, 4), (70, 8), (128, 16), (41, 16)):
        E = 108 * AL
        n = 50
        pay = rng.integers(0, 2, (n, A), dtype=np.uint8)
        rnti = rng.integers(0, 65536, n).astype(np.uint16)
        out_d = torch.zeros(n * E, dtype=torch.uint8, device="cuda")
        ctx.pdcch_encode_batch(A, E, n, torch.from_numpy(pay.reshape(-1)).cuda(), torch.from_numpy(rnti.view(np.int16)).cuda(), out_d)
        torch.cuda.synchronize()
        out = out_d.cpu().numpy().reshape(n, E)
        for i in range(n):
            assert np.array_equal(out[i], o_pdcch_encode(pay[i], int(rnti[i]), E)), (A, AL, i)


def test_polar_invalid_code(ctx):
    import miphy
    with pytest.raises(RuntimeError):
        miphy.PolarCode(30, 100, 9, 0).info()  # downlink needs 36 <= K <= 164 (polar_code_impl.cpp:335-341)
    with pytest.raises(RuntimeError):
        miphy.PolarCode(50, 40, 10, 0).info()  # E must exceed K


def test_pbch_encoder_batch(ctx):
    import torch
    import miphy
    rng = np.random.default_rng(43)
    n = 64
    msgs = np.zeros(n, dtype=miphy.PbchMsg)
    for i in range(n):
        L_max = [4, 8, 64][i % 3]
        msgs[i] = (int(rng.integers(0, 1008)), int(rng.integers(0, L_max)), L_max, int(rng.integers(0, 2)), int(rng.integers(0, 1024)),
                   int(rng.integers(0, 12 if L_max == 64 else 24)), rng.integers(0, 2, 32, dtype=np.uint8))
    out_d = torch.zeros(n * 864, dtype=torch.uint8, device="cuda")
    ctx.pbch_encode_batch(msgs, out_d)
    torch.cuda.synchronize()
    out = out_d.cpu().numpy().reshape(n, 864)
    for i in range(n):
        m = msgs[i]
        exp = o_pbch_encode(int(m["N_id"]), int(m["ssb_idx"]), int(m["L_max"]), int(m["hrf"]), int(m["sfn"]), int(m["k_ssb"]), m["payload"])
        assert np.array_equal(out[i], exp), i


@pytest.mark.parametrize("A,AL,sigma", [(12, 1, 1.1), (40, 1, 0.9), (40, 2, 1.25), (70, 4, 1.6), (128, 8, 1.5), (57, 16, 3.0)])
def test_scl_list_decoder_pdcch(ctx, A, AL, sigma):
    """SCL (L = 1, 2, 4, 8; plain and CRC-aided) against the oracle's restatement, bit for bit incl. metric and CRC verdict
    (BASELINE configs[3]: PDCCH aggregation levels 1-16). There is no reference counterpart for L > 1; properties checked on
    top: CA-SCL-8 never returns a wrong payload with crc_ok, and it decodes about as many blocks as the reference-style SSC or more."""
    import torch
    import miphy
    rng = np.random.default_rng(A * 100 + AL)
    K, E = A + 24, 108 * AL
    code = miphy.PolarCode(K, E, 9, 0)
    nb = 48
    pays = rng.integers(0, 2, (nb, A), dtype=np.uint8)
    rntis = rng.integers(0, 65536, nb).astype(np.uint16)
    llrs = np.zeros((nb, E), np.int8)
    for i in range(nb):
        tx = o_pdcch_encode(pays[i], int(rntis[i]), E)
        y = (1.0 - 2.0 * tx) + sigma * rng.standard_normal(E)
        # channel LLR 2 y / sigma^2 on the reference's int8 scale (range 20 -> 120): with a fixed gain instead, the repetition
        # combining of the high aggregation levels saturates every soft bit and no decoder has soft information left
        llrs[i] = np.round(np.clip(2 * y / sigma ** 2, -20, 20) / 20 * 120)
    l_d = torch.from_numpy(llrs.reshape(-1)).cuda()
    r_d = torch.from_numpy(rntis.view(np.int16)).cuda()
    n_ca8 = n_ssc = 0
    for L in (1, 2, 4, 8):
        for mode in (0, 1):
            msg_d = torch.zeros(nb * K, dtype=torch.uint8, device="cuda")
            ok_d = torch.zeros(nb, dtype=torch.uint8, device="cuda")
            pm_d = torch.zeros(nb, dtype=torch.int32, device="cuda")
            ctx.polar_decode_list_batch(code, L, mode, nb, l_d, r_d if mode == 1 else None, msg_d, ok_d, pm_d)
            torch.cuda.synchronize()
            msg, ok, pm = msg_d.cpu().numpy().reshape(nb, K), ok_d.cpu().numpy(), pm_d.cpu().numpy()
            for i in range(nb):
                em, eok, epm = o_polar_scl_decode(K, E, 9, 0, L, mode, int(rntis[i]), llrs[i])
                assert np.array_equal(msg[i], em) and bool(ok[i]) == eok and int(pm[i]) == epm, (L, mode, i)
                if mode == 1 and L == 8:
                    if ok[i]:
                        assert np.array_equal(msg[i][:A], pays[i]), "CRC-passing candidate with a wrong payload"
                        n_ca8 += 1
    for i in range(nb):
        m, _, _ = o_polar_decode_chain(K, E, 9, 0, llrs[i])
        import ctypes as C
        from oracle_lib import oracle
        c = np.zeros(K, np.uint8)
        oracle().orc_polar_interleave(m.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p), C.c_uint(K), 1)
        n_ssc += int(np.array_equal(c[:A], pays[i]))
    # at these operating points the list decoder recovers 1.4-3 times as many blocks as the reference-style SSC decoder
    assert n_ca8 >= n_ssc + 5, (n_ca8, n_ssc)


def test_scl_pbch_and_uplink_codes(ctx):
    """crc_mode 2 (PBCH, K = 56, E = 864) and plain list decoding of uplink codes with parity-check bits / all rate-matching modes."""
    import torch
    import miphy
    rng = np.random.default_rng(44)
    for (K, E, nMax, ibil, mode) in ((56, 864, 9, 0, 2), (20, 100, 10, 1, 0), (25, 300, 10, 0, 0), (100, 200, 10, 1, 0), (500, 1500, 10, 0, 0),
                                     (300, 400, 10, 1, 0)):
        code = miphy.PolarCode(K, E, nMax, ibil)
        nb = 8
        llrs = np.zeros((nb, E), np.int8)
        for i in range(nb):
            msg = rng.integers(0, 2, K, dtype=np.uint8)
            tx = o_polar_encode_chain(K, E, nMax, ibil, msg)[0]
            y = (1.0 - 2.0 * tx) + 0.9 * rng.standard_normal(E)
            llrs[i] = np.round(np.clip(4 * y, -20, 20) / 20 * 120)
        msg_d = torch.zeros(nb * K, dtype=torch.uint8, device="cuda")
        ok_d = torch.zeros(nb, dtype=torch.uint8, device="cuda")
        pm_d = torch.zeros(nb, dtype=torch.int32, device="cuda")
        ctx.polar_decode_list_batch(code, 8, mode, nb, torch.from_numpy(llrs.reshape(-1)).cuda(), None, msg_d, ok_d, pm_d)
        torch.cuda.synchronize()
        msg, ok, pm = msg_d.cpu().numpy().reshape(nb, K), ok_d.cpu().numpy(), pm_d.cpu().numpy()
        for i in range(nb):
            em, eok, epm = o_polar_scl_decode(K, E, nMax, ibil, 8, mode, 0, llrs[i])
            assert np.array_equal(msg[i], em) and bool(ok[i]) == eok and int(pm[i]) == epm, (K, E, i)


@pytest.mark.parametrize("K,E,nMax,ibil", polar_cases())
def test_scl_list1_is_pinned_to_the_reference_style_decoder(ctx, K, E, nMax, ibil):
    """List size 1 of polar_scl_kernel (plain mode) against (a) the successive-cancellation decoder written from the definition
    (oracle: orc_polar_sc_textbook) -- always identical -- and (b) the SSC kernel, which is pinned to the reference's decoder: identical
    on every codeword without a tie (an LLR of exactly zero at an information leaf), noiseless, AWGN and full-range inputs incl.
    +-127. The ties are the one principled divergence (tests/test_polar_sc_pinning.py); L > 1 stays parity unpinned."""
    import torch
    import miphy
    from oracle_lib import o_polar_sc_textbook
    from test_polar_sc_pinning import stimuli
    rng = np.random.default_rng(K * 13 + E + ibil)
    msgs, llrs = stimuli(K, E, nMax, ibil, rng, nb=8)
    nb = llrs.shape[0]
    code = miphy.PolarCode(K, E, nMax, ibil)
    l_d = torch.from_numpy(llrs.reshape(-1)).cuda()
    m1 = torch.zeros(nb * K, dtype=torch.uint8, device="cuda")
    ok = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    ctx.polar_decode_list_batch(code, 1, 0, nb, l_d, None, m1, ok)
    m2 = torch.zeros(nb * K, dtype=torch.uint8, device="cuda")
    ctx.polar_decode_batch(code, nb, l_d, m2)
    torch.cuda.synchronize()
    scl1, ssc = m1.cpu().numpy().reshape(nb, K), m2.cpu().numpy().reshape(nb, K)
    for i in range(nb):
        ref_sc, tie = o_polar_sc_textbook(K, E, nMax, ibil, llrs[i])
        assert np.array_equal(scl1[i], ref_sc), (K, E, i)
        if not tie:
            assert np.array_equal(scl1[i], ssc[i]), (K, E, i)
        if i < 2:
            assert np.array_equal(scl1[i], msgs[i])


@pytest.mark.parametrize("K,E,nMax,ibil", polar_cases()[::3])
def test_polar_single_blocks_match_the_chain_intermediates(ctx, K, E, nMax, ibil):
    """miphy_polar_block_batch: every block of the reference's polar chain on its own (allocator, encoder, rate matcher, rate dematcher,
    decoder, deallocator, interleaver: the classes behind polar_factory, channel_coding_factories.h:107-121) equals the intermediate
    the oracle chain exposes for the same codeword (which is pinned to the reference block by block, tests/test_oracle_vs_ref.py)."""
    import torch
    import miphy
    rng = np.random.default_rng(K * 11 + E)
    code = miphy.PolarCode(K, E, nMax, ibil)
    n_log, N, nPC = code.info()
    nb = 5
    msgs = rng.integers(0, 2, (nb, K), dtype=np.uint8)
    exp = [o_polar_encode_chain(K, E, nMax, ibil, msgs[i]) for i in range(nb)]

    def run(op, param, x, n_out, dtype=torch.uint8):
        xd = torch.from_numpy(np.ascontiguousarray(x).reshape(-1)).cuda()
        od = torch.full((nb * n_out,), 9, dtype=dtype, device="cuda")
        ctx.polar_block_batch(code, op, param, nb, xd, od)
        torch.cuda.synchronize()
        return od.cpu().numpy().reshape(nb, n_out)

    OP = miphy.binding
    alloc = run(OP.POLAR_OP_ALLOCATE, 0, msgs, N)
    assert np.array_equal(alloc, np.stack([e[1] for e in exp]))
    enc = run(OP.POLAR_OP_ENCODE, n_log, alloc, N)
    assert np.array_equal(enc, np.stack([e[2] for e in exp]))
    rm = run(OP.POLAR_OP_RATE_MATCH, 0, enc, E)
    assert np.array_equal(rm, np.stack([e[0] for e in exp]))
    # receive side on noisy LLRs with a few infinities
    llr = np.clip(np.round((1.0 - 2.0 * rm) * 40 + 25 * rng.standard_normal(rm.shape)), -120, 120).astype(np.int8)
    llr[rng.random(llr.shape) < 0.02] = 127
    dexp = [o_polar_decode_chain(K, E, nMax, ibil, llr[i]) for i in range(nb)]
    dem = run(OP.POLAR_OP_RATE_DEMATCH, 0, llr, N, torch.int8)
    assert np.array_equal(dem, np.stack([d[1] for d in dexp]))
    u = run(OP.POLAR_OP_DECODE, 0, dem, N)
    assert np.array_equal(u, np.stack([d[2] for d in dexp]))
    out = run(OP.POLAR_OP_DEALLOCATE, 0, u, K)
    assert np.array_equal(out, np.stack([d[0] for d in dexp]))
    # interleaver (CRC interleaver of the downlink chains, K <= 164): TX then RX is the identity and TX equals the oracle's pattern
    if K <= 164:
        tx = run(OP.POLAR_OP_INTERLEAVE_TX, K, msgs, K)
        assert np.array_equal(tx, np.stack([O_interleave(msgs[i], K, 0) for i in range(nb)]))
        assert np.array_equal(run(OP.POLAR_OP_INTERLEAVE_RX, K, tx, K), msgs)
