"""GPU parity: HIP LDPC decoder (through the C ABI) vs the CPU oracle -- bit-exact on hard bits and iteration count."""
import numpy as np
import pytest

from oracle_lib import (ALL_Z, BG_K, BG_NS, CRC16, CRC24A, CRC24B, o_crc_bits, o_ldpc_decode, o_ldpc_encode)

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["auto", "scalar", "packed"], autouse=True)
def ldpc_kernel(request):
    """Every test of this file runs with the automatic kernel choice and with each of the two decoder kernels forced."""
    import miphy
    miphy.lib().miphy_debug_force_ldpc_kernel({"auto": 0, "scalar": 1, "packed": 2}[request.param])
    yield request.param
    miphy.lib().miphy_debug_force_ldpc_kernel(0)


def noisy_llr(cw, sigma, rng):
    y = (1.0 - 2.0 * (cw & 1)) + sigma * rng.standard_normal(cw.size)
    return np.round(np.clip(4 * y, -20, 20) / 20 * 120).astype(np.int8)


def make_codeword(bg, Z, rng, nof_filler=0, poly=CRC24B):
    K = BG_K[bg] * Z
    nb = 16 if poly == CRC16 else 24
    msg = rng.integers(0, 2, K, dtype=np.uint8)
    c = o_crc_bits(poly, msg[:K - nof_filler - nb])
    msg[K - nof_filler - nb:K - nof_filler] = [(c >> (nb - 1 - i)) & 1 for i in range(nb)]
    if nof_filler:
        msg[K - nof_filler:] = 254
    return msg, o_ldpc_encode(bg, Z, msg, BG_NS[bg] * Z)


def run_batch(ctx, cases):
    """cases: list of dict(bg,Z,llr,crc,max_iter,nf). Runs them as ONE heterogeneous batch."""
    import torch
    import miphy
    n = len(cases)
    descs = np.zeros(n, dtype=miphy.LdpcDecDesc)
    llr_off, out_off = 0, 0
    llrs, exp = [], []
    for i, c in enumerate(cases):
        K = BG_K[c["bg"]] * c["Z"]
        descs[i] = (c["bg"], c["crc"] if c["crc"] >= 0 else miphy.CRC_NONE, c["Z"], c["max_iter"], c["nf"], c["llr"].size, 0,
                    llr_off, out_off)
        llrs.append(c["llr"])
        llr_off += c["llr"].size
        out_off += (K + 7) // 8
    llr_d = torch.from_numpy(np.concatenate(llrs)).cuda()
    out_d = torch.full((out_off,), 0x5A, dtype=torch.uint8, device="cuda")
    it_d = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    ctx.ldpc_decode_batch(descs, llr_d, out_d, it_d)
    torch.cuda.synchronize()
    out, its = out_d.cpu().numpy(), it_d.cpu().numpy()
    bad = []
    for i, c in enumerate(cases):
        K = BG_K[c["bg"]] * c["Z"]
        nb = (K + 7) // 8
        init = np.full(nb, 0x5A, dtype=np.uint8)
        ito, oo = o_ldpc_decode(c["bg"], c["Z"], c["llr"], c["nf"], c["crc"], c["max_iter"], out_init=init)
        o0 = int(descs[i]["out_offset"])
        if ito != its[i] or not np.array_equal(oo, out[o0:o0 + nb]):
            bad.append((i, c["bg"], c["Z"], c["crc"], c["max_iter"], c["nf"], c["llr"].size, ito, int(its[i]),
                        int(np.sum(oo != out[o0:o0 + nb]))))
    return bad


def test_all_graphs_noisy(ctx):
    """Every base graph x lifting size (the 102 cases of ldpc_enc_dec_test.cpp:226-320), three input lengths,
    with/without CRC early stop."""
    rng = np.random.default_rng(11)
    cases = []
    for bg in (1, 2):
        for Z in ALL_Z:
            K = BG_K[bg] * Z
            for trial in range(3):
                nf = int(rng.integers(0, max(1, Z // 2))) if trial == 1 else 0
                poly = CRC24B if K - nf > 60 else CRC16
                if K - nf <= 26:
                    nf = 0
                msg, cw = make_codeword(bg, Z, rng, nf, poly)
                L = [BG_NS[bg] * Z, K + 2 * Z, (K + 2 * Z + BG_NS[bg] * Z) // 2 // Z * Z][trial]
                llr = noisy_llr(cw[:L], [0.6, 0.3, 0.9][trial], rng)
                if nf:
                    llr[K - 2 * Z - nf:K - 2 * Z] = 127
                for crc in (poly, -1):
                    for mi in (1, 6):
                        cases.append(dict(bg=bg, Z=Z, llr=llr, crc=crc, max_iter=mi, nf=nf))
    bad = run_batch(ctx, cases)
    assert not bad, bad[:10]


def test_plus_minus_ten_and_zero(ctx):
    """ldpc_enc_dec_test.cpp: +-10 LLRs from the encoded bits decode in one iteration; all-zero LLRs give nullopt and
    all-ones output (only without a CRC)."""
    rng = np.random.default_rng(12)
    cases = []
    for bg, Z in ((1, 384), (1, 2), (2, 3), (2, 208), (1, 15), (2, 384), (1, 96)):
        K = BG_K[bg] * Z
        poly = CRC24B if K > 60 else CRC16
        msg, cw = make_codeword(bg, Z, rng, 0, poly)
        llr = (10 - 20 * (cw & 1).astype(np.int16)).astype(np.int8)
        cases.append(dict(bg=bg, Z=Z, llr=llr, crc=poly, max_iter=1, nf=0))
        cases.append(dict(bg=bg, Z=Z, llr=np.zeros_like(llr), crc=-1, max_iter=6, nf=0))
        cases.append(dict(bg=bg, Z=Z, llr=np.zeros_like(llr), crc=poly, max_iter=6, nf=0))
        # trailing zeros shorten the number of processed layers
        l2 = llr.copy()
        l2[K + 3 * Z + Z // 2:] = 0
        cases.append(dict(bg=bg, Z=Z, llr=l2, crc=poly, max_iter=3, nf=0))
    bad = run_batch(ctx, cases)
    assert not bad, bad[:10]


def test_full_range_random_llrs(ctx):
    """Arbitrary int8 inputs in [-127,127] including +-infinity: exercises clamp / promotion / infinity stickiness."""
    rng = np.random.default_rng(13)
    cases = []
    for bg, Z in ((1, 384), (2, 384), (1, 352), (2, 64), (1, 36), (2, 7), (1, 5)):
        for t in range(4):
            n = BG_NS[bg] * Z
            r = rng.integers(-127, 128, n).astype(np.int8)
            r[rng.random(n) < 0.05] = 127
            r[rng.random(n) < 0.05] = -127
            r[rng.random(n) < 0.1] = 0
            cases.append(dict(bg=bg, Z=Z, llr=r, crc=[-1, CRC24A, CRC16, CRC24B][t], max_iter=[2, 4, 6, 10][t], nf=0))
    bad = run_batch(ctx, cases)
    assert not bad, bad[:10]


def test_large_uniform_batch_device_descs(ctx):
    """BASELINE config: BG1 Z=384 rate ~0.88 codeblocks (4 layers), 6 iterations, device-resident descriptors."""
    import torch
    import miphy
    from miphy.ldpc import make_dec_descs
    rng = np.random.default_rng(14)
    bg, Z, nf = 1, 384, 0
    K, N = 22 * Z, 66 * Z
    n_unique, n = 8, 1024
    E = 8976
    llrs = np.zeros((n_unique, N), dtype=np.int8)
    msgs = []
    for u in range(n_unique):
        msg, cw = make_codeword(bg, Z, rng, nf, CRC24B)
        llrs[u, :E] = noisy_llr(cw[:E], 0.2, rng)
        msgs.append(msg)
    idx = rng.integers(0, n_unique, n)
    batch = llrs[idx]
    descs = make_dec_descs(n, bg, Z, N, miphy.CRC24B, 6, nf)
    d_descs = torch.from_numpy(descs.view(np.uint8)).cuda()
    llr_d = torch.from_numpy(batch.reshape(-1)).cuda()
    out_d = torch.zeros(n * K // 8, dtype=torch.uint8, device="cuda")
    it_d = torch.zeros(n, dtype=torch.int32, device="cuda")
    ctx.ldpc_decode_batch(d_descs, llr_d, out_d, it_d)
    torch.cuda.synchronize()
    out = out_d.cpu().numpy().reshape(n, K // 8)
    its = it_d.cpu().numpy()
    exp = [o_ldpc_decode(bg, Z, llrs[u], nf, CRC24B, 6) for u in range(n_unique)]
    for i in range(n):
        assert its[i] == exp[idx[i]][0]
        assert np.array_equal(out[i], exp[idx[i]][1])
    # decoded message equals the transmitted one
    for u in range(n_unique):
        assert exp[u][0] > 0
        assert np.array_equal(np.unpackbits(exp[u][1])[:K], msgs[u] & 1)


@pytest.mark.parametrize("bg,Z,use_limits", [(2, 208, True), (2, 208, False), (1, 208, True), (2, 384, True), (2, 6, True)])
def test_device_descriptors_any_base_graph(ctx, bg, Z, use_limits):
    """Device-resident descriptors: the host cannot see the base graph, so LDS must be sized for either
    (regression: BG2 full-length codeblocks reach 42 layers, more than BG1 at the same input length)."""
    import torch
    import miphy
    from miphy.ldpc import make_dec_descs
    rng = np.random.default_rng(15)
    K, N = BG_K[bg] * Z, BG_NS[bg] * Z
    poly = CRC24B if K > 60 else CRC16
    n = 3
    llrs, exp = [], []
    for i in range(n):
        msg, cw = make_codeword(bg, Z, rng, 0, poly)
        llr = noisy_llr(cw, 0.7, rng)
        llrs.append(llr)
        exp.append(o_ldpc_decode(bg, Z, llr, 0, poly, 6))
    descs = make_dec_descs(n, bg, Z, N, poly, 6, 0)
    d_descs = torch.from_numpy(descs.view(np.uint8)).cuda()
    out_d = torch.zeros(n * ((K + 7) // 8), dtype=torch.uint8, device="cuda")
    it_d = torch.zeros(n, dtype=torch.int32, device="cuda")
    ctx.ldpc_decode_batch(d_descs, torch.from_numpy(np.concatenate(llrs)).cuda(), out_d, it_d, limits=(Z, N) if use_limits else None)
    torch.cuda.synchronize()
    out, its = out_d.cpu().numpy().reshape(n, -1), it_d.cpu().numpy()
    for i in range(n):
        assert its[i] == exp[i][0] and np.array_equal(out[i], exp[i][1]), (bg, Z, i)


def test_many_launches_queued_back_to_back(ctx, ldpc_kernel):
    """The packed decoder hands out codeblocks through a work-queue counter that the launch itself leaves at zero (the workgroup that
    draws the last ticket clears it); the context cycles through 256 counters. 600 launches of 1 to 5 codeblocks queued without any
    synchronisation in between -- every counter reused -- must each decode their own codeblocks like the oracle."""
    import torch
    import miphy
    rng = np.random.default_rng(4242)
    bg, Z = 1, 128
    K, N = BG_K[bg] * Z, BG_NS[bg] * Z
    in_len = 26 * Z
    pool = []
    for k in range(6):
        msg, cw = make_codeword(bg, Z, rng)
        llr = noisy_llr(cw[:in_len], 0.45, rng)
        it, packed = o_ldpc_decode(bg, Z, llr, 0, CRC24B, 6)
        pool.append((llr, packed, it))
    llr_d = torch.from_numpy(np.concatenate([p[0] for p in pool])).cuda()
    nb = K // 8
    launches = 600
    sizes = rng.integers(1, 6, launches)
    picks = [rng.integers(0, len(pool), s) for s in sizes]
    out_d = torch.zeros(int(sizes.sum()) * nb, dtype=torch.uint8, device="cuda")
    it_d = torch.full((int(sizes.sum()),), -7, dtype=torch.int32, device="cuda")
    base = 0
    for l in range(launches):
        d = np.zeros(int(sizes[l]), dtype=miphy.LdpcDecDesc)
        for i, k in enumerate(picks[l]):
            d[i] = (bg, miphy.CRC24B, Z, 6, 0, in_len, 0, int(k) * in_len, (base + i) * nb)
        ctx.ldpc_decode_batch(d, llr_d, out_d, it_d[base:base + int(sizes[l])])
        base += int(sizes[l])
    torch.cuda.synchronize()
    out, its = out_d.cpu().numpy().reshape(-1, nb), it_d.cpu().numpy()
    base = 0
    for l in range(launches):
        for i, k in enumerate(picks[l]):
            _, packed, it = pool[int(k)]
            assert its[base + i] == it, (l, i)
            assert np.array_equal(out[base + i], packed[:nb]), (l, i)
        base += int(sizes[l])
