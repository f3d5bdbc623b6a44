"""GPU: Open Fronthaul IQ (de)compression kernels (block floating point and uncompressed fixed point) through the C ABI: bit-exact
against payloads and samples recorded from the reference (tests/golden/ofh_iq.npz), against the oracle on a slot's worth of
sections with mixed formats, widths and misaligned offsets, and the compress -> decompress round trip at full size."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _jobs(miphy, specs):
    j = np.zeros(len(specs), dtype=miphy.OfhIqJob)
    for i, spec in enumerate(specs):
        po, go, nprb, w = spec[:4]
        j[i] = (po, go, nprb, w, spec[4] if len(spec) > 4 else O.OFH_BFP)
    return j


def test_golden_vectors(ctx):
    import torch
    import miphy
    g = np.load(os.path.join(GOLD, "ofh_iq.npz"))
    for i, (comp, w, nprb) in enumerate(g["cases"].tolist()):
        pl = g["dec_payload_%d" % i]
        for simd, key in ((True, "dec_simd_%d"), (False, "dec_generic_%d")):
            out = torch.full((nprb * 12 + 3,), float("nan"), dtype=torch.complex64, device="cuda")
            ctx.ofh_iq_decompress_batch(_jobs(miphy, [(5, 3, nprb, w, comp)]), torch.from_numpy(np.concatenate([np.zeros(5, np.uint8), pl])).cuda(), out, simd)
            torch.cuda.synchronize()
            o = out.cpu().numpy()
            assert np.all(np.isnan(o[:3].real)) and np.array_equal(o[3:].view(np.uint32), g[key % i].view(np.uint32)), (comp, w, nprb, simd)
        if w >= 8:
            x = torch.from_numpy(g["cmp_in_%d" % i]).cuda()
            for sc in (1.0, 0.37):
                want = g["cmp_out_%d_%d" % (i, int(sc * 100))]
                pd = torch.full((want.size + 9,), 0xAB, dtype=torch.uint8, device="cuda")
                ctx.ofh_iq_compress_batch(_jobs(miphy, [(7, 0, nprb, w, comp)]), x, pd, sc)
                torch.cuda.synchronize()
                p = pd.cpu().numpy()
                assert np.all(p[:7] == 0xAB) and np.all(p[7 + want.size:] == 0xAB) and np.array_equal(p[7:7 + want.size], want), (comp, w, nprb, sc)


@pytest.mark.parametrize("on_device", [False, True])
def test_sections_of_a_slot_match_oracle(ctx, on_device):
    """56 sections (14 symbols x 4 ports) of mixed widths and sizes, payloads back to back (odd byte offsets), one grid."""
    import torch
    import miphy
    rng = np.random.default_rng(31)
    specs, po, go, payloads, want = [], 0, 0, [], []
    for s in range(56):
        w = int(rng.choice([9, 9, 9, 14, 12, 16, 8, 5]))
        comp = int(rng.choice([O.OFH_BFP, O.OFH_BFP, O.OFH_NONE]))
        nprb = int(rng.choice([273, 273, 106, 51, 1, 2, 3]))
        pl = rng.integers(0, 256, O.ofh_payload_bytes(nprb, w, comp), dtype=np.uint8)
        if comp == O.OFH_BFP:
            pl[::1 + 3 * w] = rng.integers(0, 16 - w + 1, nprb)
        specs.append((po, go, nprb, w, comp))
        payloads.append(pl)
        want.append(O.o_ofh_iq_decompress(pl, nprb, w, True, comp))
        po += pl.size
        go += nprb * 12 + int(rng.integers(0, 3))
    jobs = _jobs(miphy, specs)
    jd = torch.from_numpy(jobs.view(np.uint8).copy()).cuda() if on_device else jobs
    out = torch.zeros(go, dtype=torch.complex64, device="cuda")
    ctx.ofh_iq_decompress_batch(jd, torch.from_numpy(np.concatenate(payloads)).cuda(), out, True)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    for (p0, g0, nprb, w, comp), y in zip(specs, want):
        assert np.array_equal(o[g0:g0 + nprb * 12].view(np.uint32), y.view(np.uint32)), (nprb, w, comp)
    # compression of the same rows (widths >= 8), every record checked against the oracle
    cspecs, po = [], 0
    for (p0, g0, nprb, w, comp) in specs:
        w = max(w, 8)
        cspecs.append((po, g0, nprb, w, comp))
        po += O.ofh_payload_bytes(nprb, w, comp)
    x = np.nan_to_num(o * np.float32(0.9), nan=0.0, posinf=0.0, neginf=0.0).astype(np.complex64)
    jobs = _jobs(miphy, cspecs)
    jd = torch.from_numpy(jobs.view(np.uint8).copy()).cuda() if on_device else jobs
    pd = torch.zeros(po, dtype=torch.uint8, device="cuda")
    ctx.ofh_iq_compress_batch(jd, torch.from_numpy(x).cuda(), pd, 0.8)
    torch.cuda.synchronize()
    p = pd.cpu().numpy()
    for (p0, g0, nprb, w, comp) in cspecs:
        assert np.array_equal(p[p0:p0 + O.ofh_payload_bytes(nprb, w, comp)], O.o_ofh_iq_compress(x[g0:g0 + nprb * 12], nprb, w, 0.8, comp)), (nprb, w, comp)


def test_full_size_round_trip(ctx):
    """256 slots x 14 symbols of 273 PRB at 9 bits: decompress(compress(x)) stays within the block's quantisation step, and
    compressing the decompressed samples again reproduces the payload (idempotence)."""
    import torch
    import miphy
    n, nprb, w = 256 * 14, 273, 9
    rec = O.ofh_payload_bytes(nprb, w)
    gen = torch.Generator(device="cuda").manual_seed(3)
    amp = 10 ** (torch.rand(n * nprb, device="cuda", generator=gen) * 3 - 3.2)
    x = (torch.randn(n * nprb * 12, 2, device="cuda", generator=gen) * amp.repeat_interleave(12)[:, None]).clamp(-0.999, 0.999)
    x = torch.view_as_complex(x.contiguous())
    jobs = _jobs(miphy, [(i * rec, i * nprb * 12, nprb, w) for i in range(n)])
    p1 = torch.zeros(n * rec, dtype=torch.uint8, device="cuda")
    ctx.ofh_iq_compress_batch(jobs, x, p1)
    y = torch.zeros_like(x)
    ctx.ofh_iq_decompress_batch(jobs, p1, y)
    p2 = torch.zeros_like(p1)
    ctx.ofh_iq_compress_batch(jobs, y, p2)
    torch.cuda.synchronize()
    assert torch.equal(p1, p2)
    exps = p1.view(n * nprb, 1 + 3 * w)[:, 0].to(torch.float32)
    assert int(exps.max()) <= 7 and int(exps.min()) == 0
    step = (2.0 ** exps / 32767).repeat_interleave(12)
    err = (torch.view_as_real(y) - torch.view_as_real(x)).abs().amax(dim=1)
    assert bool((err <= step * 1.001 + 1e-7).all())


def test_full_size_uncompressed_round_trip(ctx):
    """The 16-bit uncompressed format at full size: the error stays below one quantisation step and a second compression of the
    decompressed samples reproduces the payload."""
    import torch
    import miphy
    n, nprb, w = 64 * 14, 273, 16
    rec = O.ofh_payload_bytes(nprb, w, O.OFH_NONE)
    gen = torch.Generator(device="cuda").manual_seed(4)
    x = torch.view_as_complex((torch.randn(n * nprb * 12, 2, device="cuda", generator=gen) * 0.2).clamp(-0.999, 0.999).contiguous())
    jobs = _jobs(miphy, [(i * rec, i * nprb * 12, nprb, w, O.OFH_NONE) for i in range(n)])
    p1 = torch.zeros(n * rec, dtype=torch.uint8, device="cuda")
    ctx.ofh_iq_compress_batch(jobs, x, p1)
    y = torch.zeros_like(x)
    ctx.ofh_iq_decompress_batch(jobs, p1, y)
    p2 = torch.zeros_like(p1)
    ctx.ofh_iq_compress_batch(jobs, y, p2)
    torch.cuda.synchronize()
    assert torch.equal(p1, p2)
    assert float((torch.view_as_real(y) - torch.view_as_real(x)).abs().max()) <= 0.5 / 32767 + 1e-7


def test_errors(ctx):
    import torch
    import miphy
    pl = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    g = torch.zeros(4096, dtype=torch.complex64, device="cuda")
    for spec, msg in [((0, 0, 0, 9), "PRBs out of range"), ((0, 0, 276, 9), "PRBs out of range"), ((0, 0, 4, 0), "data width"), ((0, 0, 4, 17), "data width")]:
        with pytest.raises(RuntimeError, match=msg):
            ctx.ofh_iq_decompress_batch(_jobs(miphy, [spec]), pl, g)
    with pytest.raises(RuntimeError, match="data width 7 not supported"):
        ctx.ofh_iq_compress_batch(_jobs(miphy, [(0, 0, 4, 7)]), g, pl)
    for comp in (2, 3, 6):  # block scaling, mu-law, ...: the reference's factory returns implementations that abort
        with pytest.raises(RuntimeError, match="not implemented"):
            ctx.ofh_iq_decompress_batch(_jobs(miphy, [(0, 0, 4, 9, comp)]), pl, g)
    with pytest.raises(ValueError, match="device"):
        ctx.ofh_iq_decompress_batch(_jobs(miphy, [(0, 0, 4, 9)]), pl.cpu(), g)
