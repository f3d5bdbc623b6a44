"""GPU parity for the LDPC encoder, rate matcher, rate dematcher and CRC kernels vs the CPU oracle (bit-exact)."""
import numpy as np
import pytest

from oracle_lib import (ALL_Z, BG_K, BG_NS, o_crc_bits, o_ldpc_encode, o_rate_dematch, o_rate_match)

pytestmark = pytest.mark.gpu


def test_encoder_all_graphs(ctx):
    import torch
    import miphy
    rng = np.random.default_rng(21)
    cases = []
    for bg in (1, 2):
        for Z in ALL_Z:
            K = BG_K[bg] * Z
            for ol in (BG_NS[bg] * Z, K + 2 * Z, min(BG_NS[bg] * Z, K + 5 * Z + 3)):
                msg = rng.integers(0, 2, K, dtype=np.uint8)
                nf = int(rng.integers(0, Z))
                if nf:
                    msg[-nf:] = 254
                cases.append((bg, Z, msg, ol))
    descs = np.zeros(len(cases), dtype=miphy.LdpcEncDesc)
    io, oo = 0, 0
    for i, (bg, Z, msg, ol) in enumerate(cases):
        descs[i] = (bg, 0, Z, ol, io, oo)
        io += msg.size
        oo += ol
    msg_d = torch.from_numpy(np.concatenate([c[2] for c in cases])).cuda()
    out_d = torch.full((oo,), 77, dtype=torch.uint8, device="cuda")
    ctx.ldpc_encode_batch(descs, msg_d, out_d)
    torch.cuda.synchronize()
    out = out_d.cpu().numpy()
    for i, (bg, Z, msg, ol) in enumerate(cases):
        o0 = int(descs[i]["out_offset"])
        exp = o_ldpc_encode(bg, Z, msg, ol)
        assert np.array_equal(out[o0:o0 + ol], exp), (bg, Z, ol, np.flatnonzero(out[o0:o0 + ol] != exp)[:5])


def _rm_cases(rng):
    cases = []
    for bg in (1, 2):
        for Z in (2, 3, 7, 16, 52, 104, 208, 384):
            N = BG_NS[bg] * Z
            K = BG_K[bg] * Z
            for rv in range(4):
                for mod in (1, 2, 4, 6, 8):
                    for Nref in (0, N - 3 * Z, (N * 2) // 3):
                        if Nref and Nref <= (BG_K[bg] - 2) * Z:
                            continue
                        nf = int(rng.integers(0, Z))
                        for E in (mod * int(rng.integers(1, 3 * N // mod)), mod * ((K + 7 * Z) // mod)):
                            if E > 0:
                                cases.append((bg, Z, rv, mod, Nref, nf, E))
    return cases


def test_rate_matcher(ctx):
    import torch
    import miphy
    rng = np.random.default_rng(22)
    cases = _rm_cases(rng)
    descs = np.zeros(len(cases), dtype=miphy.LdpcRdmDesc)
    cbs, io, oo = [], 0, 0
    for i, (bg, Z, rv, mod, Nref, nf, E) in enumerate(cases):
        N, K = BG_NS[bg] * Z, BG_K[bg] * Z
        cb = rng.integers(0, 2, N, dtype=np.uint8)
        if nf:
            cb[K - 2 * Z - nf:K - 2 * Z] = 254
        cbs.append(cb)
        descs[i] = (bg, rv, mod, 1, Z, nf, Nref, E, io, oo)
        io += N
        oo += E
    in_d = torch.from_numpy(np.concatenate(cbs)).cuda()
    out_d = torch.full((oo,), 99, dtype=torch.uint8, device="cuda")
    ctx.ldpc_rate_match_batch(descs, in_d, out_d)
    torch.cuda.synchronize()
    out = out_d.cpu().numpy()
    for i, (bg, Z, rv, mod, Nref, nf, E) in enumerate(cases):
        o0 = int(descs[i]["out_offset"])
        exp = o_rate_match(rv, mod, Nref, nf, cbs[i], E)
        assert np.array_equal(out[o0:o0 + E], exp), cases[i]


@pytest.mark.parametrize("new_data", [1, 0])
def test_rate_dematcher(ctx, new_data):
    import torch
    import miphy
    rng = np.random.default_rng(23 + new_data)
    cases = _rm_cases(rng)
    descs = np.zeros(len(cases), dtype=miphy.LdpcRdmDesc)
    llrs, sbs, io, oo = [], [], 0, 0
    for i, (bg, Z, rv, mod, Nref, nf, E) in enumerate(cases):
        N = BG_NS[bg] * Z
        llrs.append(rng.integers(-120, 121, E).astype(np.int8))
        sbs.append(rng.integers(-120, 121, N).astype(np.int8))
        descs[i] = (bg, rv, mod, new_data, Z, nf, Nref, E, io, oo)
        io += E
        oo += N
    in_d = torch.from_numpy(np.concatenate(llrs)).cuda()
    sb_d = torch.from_numpy(np.concatenate(sbs)).cuda()
    # several launches (the C ABI caps a batch at 65535 codeblocks; also exercises repeated staging)
    step = 700
    for a in range(0, len(cases), step):
        ctx.ldpc_rate_dematch_batch(descs[a:a + step], in_d, sb_d)
    torch.cuda.synchronize()
    out = sb_d.cpu().numpy()
    for i, (bg, Z, rv, mod, Nref, nf, E) in enumerate(cases):
        N = BG_NS[bg] * Z
        o0 = int(descs[i]["out_offset"])
        exp = o_rate_dematch(rv, mod, Nref, nf, new_data, llrs[i], sbs[i])
        got = out[o0:o0 + N]
        assert np.array_equal(got, exp), (cases[i], np.flatnonzero(got != exp)[:8], got[got != exp][:8], exp[got != exp][:8])


@pytest.mark.parametrize("new_data", [1, 0])
def test_rate_dematcher_wraparound_on_aligned_soft_buffers(ctx, new_data):
    """The wrap-around geometry on 16-byte aligned soft buffers (the HARQ pool's layout), where the dematcher works on whole output
    vectors from a de-interleaved LDS copy: retransmissions whose E bits run past the end of the circular buffer (rv 1-3), first
    transmissions longer than the buffer (up to three passes over it), limited buffers, fillers, every modulation order, input at any
    byte alignment -- against the oracle (ldpc_rate_dematcher_impl.cpp:43-254, AVX2 combine rule)."""
    import torch
    import miphy
    rng = np.random.default_rng(131 + new_data)
    cases = []
    for bg in (1, 2):
        for Z in (384, 208, 112, 48, 20):
            N = BG_NS[bg] * Z
            for rv in range(4):
                for mod in (1, 2, 4, 6, 8):
                    for Nref in (0, N - 3 * Z):
                        nf = int(rng.integers(0, Z)) if rng.integers(0, 4) else 0
                        Ncb = Nref or N
                        L = Ncb - nf
                        for E in (mod * (-(-L // mod) + int(rng.integers(0, 40))), mod * int(rng.integers(L // (2 * mod), (5 * L) // (2 * mod))), mod * (L // mod)):
                            if 0 < E <= 60000:
                                cases.append((bg, Z, rv, mod, Nref, nf, E))
    descs = np.zeros(len(cases), dtype=miphy.LdpcRdmDesc)
    llrs, sbs, io, oo = [], [], 0, 0
    pad_in = []
    for i, (bg, Z, rv, mod, Nref, nf, E) in enumerate(cases):
        N = BG_NS[bg] * Z
        Np = (N + 15) // 16 * 16
        llrs.append(rng.integers(-120, 121, E).astype(np.int8))
        sb = rng.integers(-120, 121, Np).astype(np.int8)
        sbs.append(sb)
        descs[i] = (bg, rv, mod, new_data, Z, nf, Nref, E, io, oo)
        io += E + int(rng.integers(0, 4))  # any input alignment
        pad_in.append(io)
        oo += Np
    in_h = np.zeros(io + 64, dtype=np.int8)
    for i, c in enumerate(cases):
        o = int(descs[i]["in_offset"])
        in_h[o:o + c[6]] = llrs[i]
    in_d = torch.from_numpy(in_h).cuda()
    sb_d = torch.from_numpy(np.concatenate(sbs)).cuda()
    step = 500
    for a in range(0, len(cases), step):
        ctx.ldpc_rate_dematch_batch(descs[a:a + step], in_d, sb_d)
    torch.cuda.synchronize()
    out = sb_d.cpu().numpy()
    for i, (bg, Z, rv, mod, Nref, nf, E) in enumerate(cases):
        N = BG_NS[bg] * Z
        Np = (N + 15) // 16 * 16
        o0 = int(descs[i]["out_offset"])
        exp = o_rate_dematch(rv, mod, Nref, nf, new_data, llrs[i], sbs[i][:N])
        got = out[o0:o0 + N]
        assert np.array_equal(got, exp), (cases[i], np.flatnonzero(got != exp)[:8], got[got != exp][:8], exp[got != exp][:8])
        assert np.array_equal(out[o0 + N:o0 + Np], sbs[i][N:]), ("padding behind the soft buffer touched", cases[i])


def test_crc_batch(ctx):
    import torch
    import miphy
    rng = np.random.default_rng(25)
    data = rng.integers(0, 256, 200000, dtype=np.uint8)
    bits = np.unpackbits(data)
    cases = []
    for poly in range(5):
        for nbits in (1, 7, 8, 24, 31, 32, 33, 100, 1001, 8424, 8423, 319784 + 24, 1277992):
            off = int(rng.integers(0, 200000 * 8 - nbits - 64))
            cases.append((off, nbits, poly))
    descs = np.zeros(len(cases), dtype=miphy.CrcDesc)
    for i, c in enumerate(cases):
        descs[i] = c
    d_d = torch.from_numpy(np.concatenate([data, np.zeros(16, dtype=np.uint8)])).cuda()
    out_d = torch.zeros(len(cases), dtype=torch.int32, device="cuda")
    ctx.crc_batch(descs, d_d, out_d)
    torch.cuda.synchronize()
    out = out_d.cpu().numpy().astype(np.uint32)
    for i, (off, nbits, poly) in enumerate(cases):
        assert int(out[i]) == o_crc_bits(poly, bits[off:off + nbits]), cases[i]


def test_host_descriptors_queued_without_synchronisation_across_the_staging_ring(ctx):
    """Host descriptor arrays are copied at the call into the context's pinned staging ring and the call returns without waiting for
    the stream (miphy.h, execution model). 2600 calls queued back to back, each with its own descriptors (reused host array, 200
    descriptors = 3.2 KB (one 3.25 KB ring slot) per call: 8.7 MB in all, the 8 MB ring wraps inside the run), one synchronisation at the end: every call
    must have seen ITS descriptors."""
    import torch
    import miphy
    rng = np.random.default_rng(77)
    data = rng.integers(0, 256, 4096, dtype=np.uint8)
    bits = np.unpackbits(data)
    d_d = torch.from_numpy(np.concatenate([data, np.zeros(16, dtype=np.uint8)])).cuda()
    ncall, per = 2600, 200
    out_d = torch.zeros(ncall * per, dtype=torch.int32, device="cuda")
    descs = np.zeros(per, dtype=miphy.CrcDesc)  # ONE host array, overwritten for every call
    offs = rng.integers(0, 4096 * 8 - 64, (ncall, per))
    lens = rng.integers(1, 64, (ncall, per))
    for c in range(ncall):
        descs["bit_offset"], descs["nbits"], descs["poly"] = offs[c], lens[c], miphy.CRC16
        ctx.crc_batch(descs, d_d, out_d[c * per:(c + 1) * per])
    torch.cuda.synchronize()
    out = out_d.cpu().numpy().astype(np.uint32).reshape(ncall, per)
    for c in list(range(0, ncall, 97)) + [ncall - 1]:
        for i in range(0, per, 13):
            o, n = int(offs[c, i]), int(lens[c, i])
            assert int(out[c, i]) == o_crc_bits(miphy.CRC16, bits[o:o + n]), (c, i)
