"""GPU parity: HIP LDPC decoder (through the C ABI) vs the CPU oracle -- bit-exact on hard bits and iteration count."""
import numpy as np
import pytest

from oracle_lib import (ALL_Z, BG_K, BG_NS, CRC16, CRC24A, CRC24B, o_crc_bits, o_ldpc_decode, o_ldpc_encode)

pytestmark = pytest.mark.gpu


SCALAR, PACKED, FUSED, GMSG, WAVE, SPLIT, GMSG_PART = 1, 2, 4, 8, 16, 32, 64  # MIPHY_LDPC_KERNEL_* (include/miphy.h)


@pytest.fixture(params=["auto", "scalar", "packed", "throughput", "latency2"], autouse=True)
def ldpc_kernel(request):
    """Every test of this file runs with the automatic choice (host descriptors: class-sorted launches -- the wave kernel for Z <= 64,
    the packed kernel above, in its latency form because these batches hold fewer codeblocks than the chip has CUs; device descriptors:
    one launch), with the one-row-per-lane kernel forced, with the packed kernel forced as ONE launch for the whole batch, and with the
    class-sorted launches in their throughput form (what a batch that fills the chip gets), and with the latency form cut in two parts
    instead of four. `kernels_used()` tells which kernels really ran; the tests assert it."""
    import miphy
    miphy.lib().miphy_debug_force_ldpc_kernel({"auto": 0, "scalar": 1, "packed": 2, "throughput": 4, "latency2": 6}[request.param])
    miphy.lib().miphy_debug_ldpc_kernels_used(1)
    yield request.param
    miphy.lib().miphy_debug_force_ldpc_kernel(0)


def kernels_used():
    import miphy
    return int(miphy.lib().miphy_debug_ldpc_kernels_used(1))


def check_kernels(mode, used, cases):
    """The forced kernel is the one that ran; the automatic choice never takes the one-row-per-lane kernel for host descriptors, runs the
    wave kernel exactly when the batch holds Z <= 64 and the packed kernel exactly when it holds Z > 64."""
    if mode == "scalar":
        assert used == SCALAR, used
    elif mode == "packed":
        assert used & PACKED and not used & (SCALAR | WAVE | SPLIT), used
    else:
        assert not used & SCALAR, used
        if mode == "throughput":
            assert not used & SPLIT, used
        elif any(c["Z"] > 64 and c["llr"].size // c["Z"] < 40 for c in cases):
            assert used & SPLIT, used  # small batches: the latency form wherever its LDS image fits a CU
        assert bool(used & WAVE) == any(c["Z"] <= 64 for c in cases), used
        assert bool(used & PACKED) == any(c["Z"] > 64 for c in cases), used


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


def run_batch(ctx, cases, mode=None, pad=0):
    """cases: list of dict(bg,Z,llr,crc,max_iter,nf). Runs them as ONE batch with host descriptors; `pad` LLRs of garbage in front of
    every codeblock (unaligned inputs). Returns the mismatches against the oracle."""
    import torch
    import miphy
    n = len(cases)
    descs = np.zeros(n, dtype=miphy.LdpcDecDesc)
    llr_off, out_off = 0, 0
    llrs, exp = [], []
    for i, c in enumerate(cases):
        K = BG_K[c["bg"]] * c["Z"]
        if pad:
            llrs.append(np.full(pad, 77, np.int8))
            llr_off += pad
        descs[i] = (c["bg"], c["crc"] if c["crc"] >= 0 else miphy.CRC_NONE, c["Z"], c["max_iter"], c["nf"], c["llr"].size, c.get("flags", 0),
                    llr_off, out_off)
        llrs.append(c["llr"])
        llr_off += c["llr"].size
        out_off += (K + 7) // 8
    llr_d = torch.from_numpy(np.concatenate(llrs)).cuda()
    out_d = torch.full((out_off,), 0x5A, dtype=torch.uint8, device="cuda")
    it_d = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    kernels_used()
    ctx.ldpc_decode_batch(descs, llr_d, out_d, it_d)
    torch.cuda.synchronize()
    if mode is not None:
        check_kernels(mode, kernels_used(), cases)
    out, its = out_d.cpu().numpy(), it_d.cpu().numpy()
    bad = []
    for i, c in enumerate(cases):
        K = BG_K[c["bg"]] * c["Z"]
        nb = (K + 7) // 8
        init = np.full(nb, 0x5A, dtype=np.uint8)
        if c.get("flags", 0) & 1:  # CRC checked once after the last iteration (pusch_decoder_impl.cpp:105-118)
            _, oo = o_ldpc_decode(c["bg"], c["Z"], c["llr"], c["nf"], -1, c["max_iter"], out_init=init)
            if not np.any(c["llr"]):
                ito, oo = 0, init  # all-zero input with a CRC: nullopt, output untouched
            else:
                L = K - c["nf"]
                ito = c["max_iter"] if o_crc_bits(c["crc"], np.unpackbits(oo)[:L]) == 0 else 0
        else:
            ito, oo = o_ldpc_decode(c["bg"], c["Z"], c["llr"], c["nf"], c["crc"], c["max_iter"], out_init=init)
        o0 = int(descs[i]["out_offset"])
        if ito != its[i] or not np.array_equal(oo, out[o0:o0 + nb]):
            bad.append((i, c["bg"], c["Z"], c["crc"], c["max_iter"], c["nf"], c["llr"].size, ito, int(its[i]),
                        int(np.sum(oo != out[o0:o0 + nb]))))
    return bad


def run_per_size(ctx, cases, mode):
    """One batch per (base graph, lifting size): the packed kernel then runs with the workgroup size of that lifting size (1, 2 or 3
    wavefronts) and the message placement (LDS / global memory) its code rates select, instead of the geometry of the batch's largest."""
    bad, used = [], 0
    for key in sorted({(c["bg"], c["Z"]) for c in cases}):
        sub = [c for c in cases if (c["bg"], c["Z"]) == key]
        bad += run_batch(ctx, sub, mode)
    return bad


def graph_cases(rng, sizes):
    cases = []
    for bg in (1, 2):
        for Z in sizes:
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
                cases.append(dict(bg=bg, Z=Z, llr=llr, crc=poly, max_iter=4, nf=nf, flags=1))  # CRC after the last iteration only
    return cases


def test_all_graphs_noisy(ctx, ldpc_kernel):
    """Every base graph x lifting size (the 102 cases of ldpc_enc_dec_test.cpp:226-320), three input lengths (full, shortest, between),
    fillers, CRC early stop / no CRC / CRC after the last iteration -- as ONE heterogeneous batch (sizes 2 ... 384 side by side, odd ones
    included: the packed kernel then runs with most lanes of a small codeblock idle) and as one batch per lifting size."""
    rng = np.random.default_rng(11)
    cases = graph_cases(rng, ALL_Z)
    bad = run_batch(ctx, cases, ldpc_kernel)
    assert not bad, bad[:10]
    bad = run_per_size(ctx, cases, ldpc_kernel)
    assert not bad, bad[:10]
    bad = run_batch(ctx, cases[::7], ldpc_kernel, pad=5)  # codeblocks that start at any byte alignment
    assert not bad, bad[:10]


def test_message_placement_of_the_packed_kernel(ctx, ldpc_kernel):
    """The packed kernel keeps its check-to-variable messages in LDS at high code rates and in global memory (GMSG instance) where that
    keeps more codeblocks per CU: both instances must be reached and agree with the oracle."""
    if ldpc_kernel in ("scalar", "auto", "latency2"):
        return  # nothing to place: the one-row-per-lane kernel keeps compressed check-node state, the latency form always uses LDS
    rng = np.random.default_rng(17)
    seen = set()
    for bg, Z, nodes in ((2, 384, 12), (1, 384, 24), (1, 384, 66), (2, 384, 50), (1, 256, 66), (2, 128, 50), (1, 128, 24), (1, 72, 66)):
        cases = []
        for t in range(5):
            msg, cw = make_codeword(bg, Z, rng)
            llr = noisy_llr(cw[:nodes * Z], 0.55, rng)
            cases.append(dict(bg=bg, Z=Z, llr=llr, crc=CRC24B, max_iter=6, nf=0))
        kernels_used()
        bad = run_batch(ctx, cases)
        used = kernels_used()
        assert not bad, (bg, Z, nodes, bad[:5])
        assert used & PACKED
        seen.add(used & GMSG)
        if (bg, Z, nodes) == (2, 384, 12):
            assert not used & GMSG  # 19 KB of LDS per codeblock: more fit a CU than the register file takes
        if Z == 384 and nodes > 24:
            assert used & GMSG
    assert seen == {0, GMSG}


def test_messages_split_between_lds_and_global_memory(ctx, ldpc_kernel):
    """A launch class with more codeblocks than stay resident with their messages in LDS moves messages to global memory -- only the
    layers behind a boundary (the first layers keep theirs in LDS). Needs a batch that fills the chip: 640 codeblocks of BG1 / Z = 384
    at 15 and at 28 layers and of BG2 at 30 layers, eight distinct codewords each (so the oracle decodes eight). The automatic choice
    (split) and the same with every message in global memory (mode | 0x100) must both take the GMSG instance and agree with the oracle."""
    if ldpc_kernel != "auto":
        return
    import torch
    import miphy
    rng = np.random.default_rng(23)
    for bg, Z, nodes, crc in ((1, 384, 37, CRC24B), (1, 384, 50, -1), (2, 384, 40, CRC24B)):
        base = []
        for t in range(8):
            msg, cw = make_codeword(bg, Z, rng)
            base.append(noisy_llr(cw[:nodes * Z], 0.62, rng))
        n, K = 640, BG_K[bg] * Z
        nb = (K + 7) // 8
        descs = np.zeros(n, dtype=miphy.LdpcDecDesc)
        for i in range(n):
            descs[i] = (bg, crc if crc >= 0 else miphy.CRC_NONE, Z, 5, 0, nodes * Z, 0, i * nodes * Z, i * nb)
        llr_d = torch.from_numpy(np.concatenate([base[i % 8] for i in range(n)])).cuda()
        exp = [o_ldpc_decode(bg, Z, base[t], 0, crc, 5, out_init=np.full(nb, 0x5A, dtype=np.uint8)) for t in range(8)]
        for mode in (0, 0x100):
            miphy.lib().miphy_debug_force_ldpc_kernel(mode)
            out_d = torch.full((n * nb,), 0x5A, dtype=torch.uint8, device="cuda")
            it_d = torch.full((n,), -7, dtype=torch.int32, device="cuda")
            kernels_used()
            ctx.ldpc_decode_batch(descs, llr_d, out_d, it_d)
            torch.cuda.synchronize()
            used = kernels_used()
            assert used & PACKED and used & GMSG and not used & (SPLIT | SCALAR | WAVE), (bg, nodes, mode, used)
            assert bool(used & GMSG_PART) == (mode == 0), (bg, nodes, mode, used)  # the split really happened / was really switched off
            out, its = out_d.cpu().numpy().reshape(n, nb), it_d.cpu().numpy()
            for i in range(n):
                ito, oo = exp[i % 8]
                assert its[i] == ito and np.array_equal(out[i], oo), (bg, nodes, mode, i, its[i], ito)
        miphy.lib().miphy_debug_force_ldpc_kernel(0)


def corner_cases(rng, sizes):
    cases = []
    for bg, Z in sizes:
        K = BG_K[bg] * Z
        poly = CRC24B if K > 60 else CRC16
        msg, cw = make_codeword(bg, Z, rng, 0, poly)
        llr = (10 - 20 * (cw & 1).astype(np.int16)).astype(np.int8)
        cases.append(dict(bg=bg, Z=Z, llr=llr, crc=poly, max_iter=1, nf=0))
        cases.append(dict(bg=bg, Z=Z, llr=np.zeros_like(llr), crc=-1, max_iter=6, nf=0))
        cases.append(dict(bg=bg, Z=Z, llr=np.zeros_like(llr), crc=poly, max_iter=6, nf=0))
        cases.append(dict(bg=bg, Z=Z, llr=np.zeros_like(llr), crc=poly, max_iter=6, nf=0, flags=1))
        # trailing zeros shorten the number of processed layers (ldpc_decoder_impl.cpp:86-114)
        for cut in (K + 3 * Z + Z // 2, K + 9 * Z + 1, BG_NS[bg] * Z - 1):
            l2 = llr.copy()
            l2[cut:] = 0
            cases.append(dict(bg=bg, Z=Z, llr=l2, crc=poly, max_iter=3, nf=0))
        # +-infinity: saturated inputs (every soft bit infinite), and a codeword whose systematic part is infinite
        cases.append(dict(bg=bg, Z=Z, llr=(127 - 254 * (cw & 1).astype(np.int16)).astype(np.int8), crc=poly, max_iter=2, nf=0))
        l3 = noisy_llr(cw, 0.8, rng)
        l3[:K - 2 * Z] = (127 - 254 * (cw[:K - 2 * Z] & 1).astype(np.int16)).astype(np.int8)
        cases.append(dict(bg=bg, Z=Z, llr=l3, crc=poly, max_iter=4, nf=0))
        l4 = noisy_llr(cw, 0.5, rng)
        l4[rng.random(l4.size) < 0.2] = -127  # wrong-signed infinities: sticky through every iteration
        cases.append(dict(bg=bg, Z=Z, llr=l4, crc=poly, max_iter=5, nf=0))
    return cases


EVEN_LARGE = [(1, 384), (2, 384), (1, 352), (2, 320), (1, 288), (2, 256), (1, 240), (2, 208), (1, 192), (2, 176), (1, 160), (2, 144), (1, 128),
              (2, 112), (1, 96), (2, 80), (1, 72)]
SMALL_AND_ODD = [(1, 2), (2, 2), (2, 3), (1, 3), (1, 5), (2, 7), (1, 9), (2, 11), (1, 13), (1, 15), (2, 15), (1, 16), (2, 22), (1, 30), (2, 36),
                 (1, 44), (2, 52), (1, 60), (2, 64), (1, 64)]


@pytest.mark.parametrize("sizes", ["even_large", "small_and_odd"])
def test_plus_minus_ten_zero_tail_and_infinities(ctx, ldpc_kernel, sizes):
    """ldpc_enc_dec_test.cpp: +-10 LLRs from the encoded bits decode in one iteration; all-zero LLRs give nullopt and all-ones output
    (only without a CRC); zero tails cut the layer loop (ldpc_decoder_impl.cpp:86-114); +-127 inputs run the infinity rule of
    ldpc_decoder_avx2.cpp:85-105,205-243. Every even lifting size above 64 -- the packed kernel's domain -- and the small / odd ones
    (wave kernel), per lifting size and as one batch."""
    rng = np.random.default_rng(12)
    cases = corner_cases(rng, EVEN_LARGE if sizes == "even_large" else SMALL_AND_ODD)
    bad = run_per_size(ctx, cases, ldpc_kernel)
    assert not bad, bad[:10]
    bad = run_batch(ctx, cases, ldpc_kernel)
    assert not bad, bad[:10]


@pytest.mark.parametrize("sizes", ["even_large", "small_and_odd"])
def test_full_range_random_llrs(ctx, ldpc_kernel, sizes):
    """Arbitrary int8 inputs in [-127,127] including +-infinity: exercises clamp / promotion / infinity stickiness."""
    rng = np.random.default_rng(13)
    cases = []
    for bg, Z in (EVEN_LARGE[:10] if sizes == "even_large" else SMALL_AND_ODD):
        for t in range(4):
            n = BG_NS[bg] * Z if t != 2 else (BG_K[bg] + 7) * Z
            r = rng.integers(-127, 128, n).astype(np.int8)
            r[rng.random(n) < 0.05] = 127
            r[rng.random(n) < 0.05] = -127
            r[rng.random(n) < 0.1] = 0
            K = BG_K[bg] * Z
            crc = [-1, CRC24A, CRC16, CRC24B][t] if K > 60 else [-1, CRC16, CRC16, -1][t]
            cases.append(dict(bg=bg, Z=Z, llr=r, crc=crc, max_iter=[2, 4, 6, 10][t], nf=0))
    bad = run_per_size(ctx, cases, ldpc_kernel)
    assert not bad, bad[:10]
    bad = run_batch(ctx, cases, ldpc_kernel)
    assert not bad, bad[:10]


def test_wave_kernel_full_bundles(ctx, ldpc_kernel):
    """Z <= 64: a wavefront decodes a bundle of floor(64 / ceil(Z / 2)) codeblocks of one lifting size. Many codeblocks per size (full
    bundles and a ragged last one) that differ in input length, fillers, noise (so their early stops fall in different iterations),
    with all-zero and saturated inputs among them."""
    rng = np.random.default_rng(19)
    cases = []
    for bg, Z, count in ((1, 2, 150), (2, 3, 70), (1, 5, 45), (2, 7, 40), (1, 15, 27), (2, 16, 27), (1, 22, 14), (2, 30, 14), (1, 32, 11),
                         (2, 36, 11), (1, 44, 7), (2, 52, 7), (1, 60, 5), (2, 64, 5)):
        K = BG_K[bg] * Z
        poly = CRC24B if K > 60 else CRC16
        for i in range(count):
            nf = int(rng.integers(0, Z // 2 + 1)) if (i % 3 == 1 and K - Z // 2 > 30) else 0
            msg, cw = make_codeword(bg, Z, rng, nf, poly)
            nodes = int(rng.integers(BG_K[bg] + 2, BG_NS[bg] + 1))
            llr = noisy_llr(cw[:nodes * Z], float(rng.choice([0.2, 0.5, 0.8, 1.1])), rng)
            if nf:
                llr[K - 2 * Z - nf:K - 2 * Z] = 127
            if i % 11 == 5:
                llr[:] = 0
            if i % 13 == 7:
                llr = (127 - 254 * (cw[:nodes * Z] & 1).astype(np.int16)).astype(np.int8)
            cases.append(dict(bg=bg, Z=Z, llr=llr, crc=poly, max_iter=6, nf=nf))
    bad = run_batch(ctx, cases, ldpc_kernel, pad=1)
    assert not bad, bad[:10]
    rng.shuffle(cases)
    for c in cases[::3]:
        c["flags"] = 1
    bad = run_batch(ctx, cases[:200], ldpc_kernel)
    assert not bad, bad[:10]


def test_mixed_odd_and_even_device_descriptors(ctx, ldpc_kernel):
    """Device-resident descriptors that mix odd lifting sizes (small transport blocks) with Z = 384 under limits that only name the
    largest one: ONE launch of the packed kernel, in which an odd-Z codeblock folds its unpaired last row onto itself."""
    import torch
    import miphy
    rng = np.random.default_rng(23)
    cases = []
    for bg, Z in ((1, 384), (2, 3), (1, 5), (2, 7), (1, 9), (2, 11), (1, 13), (2, 15), (1, 15), (2, 384), (1, 3), (2, 208)):
        K = BG_K[bg] * Z
        poly = CRC24B if K > 60 else CRC16
        for t in range(2):
            msg, cw = make_codeword(bg, Z, rng, 0, poly)
            nodes = BG_NS[bg] if t == 0 else BG_K[bg] + 4
            cases.append(dict(bg=bg, Z=Z, llr=noisy_llr(cw[:nodes * Z], 0.6, rng), crc=poly if t == 0 else -1, max_iter=5, nf=0))
    n = len(cases)
    descs = np.zeros(n, dtype=miphy.LdpcDecDesc)
    llr_off, out_off = 0, 0
    for i, c in enumerate(cases):
        descs[i] = (c["bg"], c["crc"] if c["crc"] >= 0 else miphy.CRC_NONE, c["Z"], c["max_iter"], 0, c["llr"].size, 0, llr_off, out_off)
        llr_off += c["llr"].size
        out_off += (BG_K[c["bg"]] * c["Z"] + 7) // 8
    d_descs = torch.from_numpy(descs.view(np.uint8)).cuda()
    llr_d = torch.from_numpy(np.concatenate([c["llr"] for c in cases])).cuda()
    out_d = torch.full((out_off,), 0x5A, dtype=torch.uint8, device="cuda")
    it_d = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    n_max = max((c["llr"].size + 2 * c["Z"] + c["Z"] - 1) // c["Z"] for c in cases)
    kernels_used()
    ctx.ldpc_decode_batch(d_descs, llr_d, out_d, it_d, limits=(384, (n_max - 2) * 384))
    torch.cuda.synchronize()
    used = kernels_used()
    assert used == (SCALAR if ldpc_kernel == "scalar" else PACKED | (used & GMSG)), used
    out, its = out_d.cpu().numpy(), it_d.cpu().numpy()
    for i, c in enumerate(cases):
        nb = (BG_K[c["bg"]] * c["Z"] + 7) // 8
        ito, oo = o_ldpc_decode(c["bg"], c["Z"], c["llr"], 0, c["crc"], c["max_iter"], out_init=np.full(nb, 0x5A, np.uint8))
        o0 = int(descs[i]["out_offset"])
        assert ito == its[i] and np.array_equal(oo, out[o0:o0 + nb]), (i, c["bg"], c["Z"], ito, int(its[i]))


def test_large_uniform_batch_device_descs(ctx, ldpc_kernel):
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
    in_len = (E + Z - 1) // Z * Z  # 24 nodes of input: the four layers of the headline codeblock
    descs = make_dec_descs(n, bg, Z, in_len, miphy.CRC24B, 6, nf, llr_stride=N)
    d_descs = torch.from_numpy(descs.view(np.uint8)).cuda()
    llr_d = torch.from_numpy(batch.reshape(-1)).cuda()
    out_d = torch.zeros(n * K // 8, dtype=torch.uint8, device="cuda")
    it_d = torch.zeros(n, dtype=torch.int32, device="cuda")
    kernels_used()
    ctx.ldpc_decode_batch(d_descs, llr_d, out_d, it_d, limits=(Z, in_len))
    torch.cuda.synchronize()
    used = kernels_used()
    assert (used == SCALAR) if ldpc_kernel == "scalar" else (used & PACKED and not used & (SCALAR | WAVE)), used
    out = out_d.cpu().numpy().reshape(n, K // 8)
    its = it_d.cpu().numpy()
    exp = [o_ldpc_decode(bg, Z, llrs[u][:in_len], nf, CRC24B, 6) for u in range(n_unique)]
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


def test_prepared_plan_matches_the_batch_call(ctx, ldpc_kernel):
    """miphy_ldpc_decode_plan_*: the class-sorted launches of a heterogeneous batch prepared once and run twice (the second time on
    other inputs) give the oracle's results; the plan reports one launch per class."""
    import torch
    import miphy
    rng = np.random.default_rng(29)
    cases = graph_cases(rng, (2, 7, 15, 16, 36, 64, 72, 128, 144, 256, 352, 384))
    rng.shuffle(cases)
    n = len(cases)
    descs = np.zeros(n, dtype=miphy.LdpcDecDesc)
    llr_off, out_off = 0, 0
    for i, c in enumerate(cases):
        descs[i] = (c["bg"], c["crc"] if c["crc"] >= 0 else miphy.CRC_NONE, c["Z"], c["max_iter"], c["nf"], c["llr"].size, c.get("flags", 0), llr_off, out_off)
        llr_off += c["llr"].size
        out_off += (BG_K[c["bg"]] * c["Z"] + 7) // 8
    plan = miphy.LdpcDecodePlan(ctx, descs)
    assert 2 <= plan.nof_launches() <= 60
    for rep in range(2):
        if rep == 1:  # same geometry, other soft bits: sign flips keep the zero / infinity structure of every case
            for c in cases:
                flip = rng.random(c["llr"].size) < 0.03
                c["llr"] = np.where(flip, -c["llr"], c["llr"]).astype(np.int8)
        llr_d = torch.from_numpy(np.concatenate([c["llr"] for c in cases])).cuda()
        out_d = torch.full((out_off,), 0x5A, dtype=torch.uint8, device="cuda")
        it_d = torch.full((n,), -7, dtype=torch.int32, device="cuda")
        kernels_used()
        plan.run(llr_d, out_d, it_d)
        torch.cuda.synchronize()
        used = kernels_used()
        assert (used == SCALAR) if ldpc_kernel == "scalar" else (used & PACKED and used & WAVE and not used & SCALAR), used
        assert bool(used & SPLIT) == (ldpc_kernel in ("auto", "packed", "latency2"))  # (the plan is class-sorted whatever single-launch kernel is forced)
        out, its = out_d.cpu().numpy(), it_d.cpu().numpy()
        for i, c in enumerate(cases):
            if c.get("flags", 0) & 1:
                continue  # covered by run_batch
            nb = (BG_K[c["bg"]] * c["Z"] + 7) // 8
            ito, oo = o_ldpc_decode(c["bg"], c["Z"], c["llr"], c["nf"], c["crc"], c["max_iter"], out_init=np.full(nb, 0x5A, np.uint8))
            o0 = int(descs[i]["out_offset"])
            assert ito == its[i] and np.array_equal(oo, out[o0:o0 + nb]), (rep, i, c["bg"], c["Z"], ito, int(its[i]))
    plan.close()
