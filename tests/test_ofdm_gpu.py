"""GPU parity for the batched DFT and the OFDM slot (de)modulator vs the double-precision oracle.

Tolerance (floating point): max |err| <= 4e-6 * rms(output) -- the reference's own float radix-2 DFT sits at ~1.1e-6 * rms
from the exact transform (measured in tests/test_oracle_vs_ref.py); the reference's vector tests allow 1e-4 absolute at
unit scale (ofdm_demodulator_vectortest.cpp:29-83) and MSE < 1e-6 (dft_processor_test.cpp:40-42)."""
import numpy as np
import pytest

from oracle_lib import OfdmCfg, o_dft, o_ofdm_demod_slot, o_ofdm_mod_slot, o_ofdm_slot_size

pytestmark = pytest.mark.gpu
TOL = 4e-6


def rel_err(a, b):
    return float(np.abs(a - b).max() / np.sqrt(np.mean(np.abs(b) ** 2)))


@pytest.mark.parametrize("N", [128, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096])
def test_dft_sizes(ctx, N):
    import torch
    rng = np.random.default_rng(N)
    n = 5
    x = (rng.uniform(-1, 1, (n, N)) + 1j * rng.uniform(-1, 1, (n, N))).astype(np.complex64)
    x_d = torch.from_numpy(x).cuda()
    for inv in (False, True):
        out_d = torch.zeros_like(x_d)
        ctx.dft_batch(N, inv, n, x_d, out_d)
        torch.cuda.synchronize()
        out = out_d.cpu().numpy()
        for i in range(n):
            assert rel_err(out[i], o_dft(x[i], inv)) < TOL, (N, inv, i)


@pytest.mark.parametrize("N", [4608, 6144, 9216, 12288, 18432, 24576, 36864, 49152])
def test_dft_large_sizes_four_step(ctx, N):
    """The remaining sizes of the reference's list (dft_processor_generic_impl.cpp:193-210) go through the four-step path.
    Checked against numpy's double-precision FFT (the O(N^2) oracle is too slow here) with the same tolerance."""
    import torch
    rng = np.random.default_rng(N)
    n = 3
    x = (rng.uniform(-1, 1, (n, N)) + 1j * rng.uniform(-1, 1, (n, N))).astype(np.complex64)
    x_d = torch.from_numpy(x).cuda()
    for inv in (False, True):
        out_d = torch.zeros_like(x_d)
        ctx.dft_batch(N, inv, n, x_d, out_d)
        torch.cuda.synchronize()
        out = out_d.cpu().numpy()
        ref = (np.fft.ifft(x.astype(np.complex128), axis=1) * N) if inv else np.fft.fft(x.astype(np.complex128), axis=1)
        for i in range(n):
            assert rel_err(out[i], ref[i]) < TOL, (N, inv, i, rel_err(out[i], ref[i]))
    # small-case cross-check of numpy against the oracle so that the substitution above is itself pinned
    y = x[0, :384]
    assert rel_err(np.fft.fft(y.astype(np.complex128)).astype(np.complex64), o_dft(y, False)) < 1e-6


def test_dft_unsupported_size(ctx):
    import torch
    x = torch.zeros(5000, dtype=torch.complex64, device="cuda")
    with pytest.raises(RuntimeError):
        ctx.dft_batch(5000, False, 1, x, x.clone())


CASES = [(1, 273, 4096, 144, 3.5e9), (1, 273, 4096, 0, 3.5e9), (1, 106, 2048, 72, 3.5e9), (0, 52, 1024, 10, 2.6e9),
         (1, 51, 1536, 0, 3.45e9), (2, 66, 1024, 36, 28e9)]


@pytest.mark.parametrize("mu,rb,N,wo,fc", CASES)
def test_ofdm_demodulate_and_modulate(ctx, mu, rb, N, wo, fc):
    import torch
    import miphy
    rng = np.random.default_rng(N + rb)
    nslots = 1 << mu
    cfg = miphy.OfdmConfig(mu, rb, N, wo, 0.5, 0.0, fc)
    ocfg = OfdmCfg(mu, rb, N, wo, 0.5, fc)
    sizes = [cfg.slot_size(s) for s in range(nslots)]
    assert sizes == [o_ofdm_slot_size(ocfg, s) for s in range(nslots)]
    nports = 2
    jobs = np.zeros(nslots * nports, dtype=miphy.OfdmJob)
    xs, off = [], 0
    for s in range(nslots):
        for p in range(nports):
            x = ((rng.standard_normal(sizes[s]) + 1j * rng.standard_normal(sizes[s])) * 0.7).astype(np.complex64)
            jobs[s * nports + p] = (off, (s * nports + p) * 14 * rb * 12, s, 0)
            xs.append(x)
            off += sizes[s]
    x_d = torch.from_numpy(np.concatenate(xs)).cuda()
    grid_d = torch.zeros(nslots * nports * 14 * rb * 12, dtype=torch.complex64, device="cuda")
    ctx.ofdm_demodulate_slots(cfg, jobs, x_d, grid_d)
    torch.cuda.synchronize()
    grid = grid_d.cpu().numpy().reshape(nslots * nports, 14, rb * 12)
    for j in range(nslots * nports):
        exp = o_ofdm_demod_slot(ocfg, j // nports, xs[j])
        assert rel_err(grid[j], exp) < TOL, (j, rel_err(grid[j], exp))
    # modulator: grid -> time, then the round trip mod -> demod recovers the grid up to the scale product
    mcfg = miphy.OfdmConfig(mu, rb, N, 0, 0.01, 0.0, fc)
    mocfg = OfdmCfg(mu, rb, N, 0, 0.01, fc)
    g = (rng.standard_normal((nslots * nports, 14, rb * 12)) + 1j * rng.standard_normal((nslots * nports, 14, rb * 12))).astype(np.complex64)
    g_d = torch.from_numpy(g.reshape(-1)).cuda()
    y_d = torch.zeros_like(x_d)
    jobs["grid_empty"][-1] = 1
    ctx.ofdm_modulate_slots(mcfg, jobs, g_d, y_d)
    torch.cuda.synchronize()
    y = y_d.cpu().numpy()
    for j in range(nslots * nports):
        o0 = int(jobs[j]["samples_offset"])
        got = y[o0:o0 + sizes[j // nports]]
        if j == nslots * nports - 1:
            assert not got.any()
            continue
        exp = o_ofdm_mod_slot(mocfg, j // nports, g[j])
        assert rel_err(got, exp) < TOL, (j, rel_err(got, exp))
    # round trip (size independent property): demod(mod(grid)) == grid * N * scale_tx * scale_rx when both use offset 0
    jobs["grid_empty"][-1] = 0
    rcfg = miphy.OfdmConfig(mu, rb, N, 0, 1.0 / (N * 0.01), 0.0, fc)
    ctx.ofdm_modulate_slots(mcfg, jobs, g_d, y_d)
    g2_d = torch.zeros_like(g_d)
    ctx.ofdm_demodulate_slots(rcfg, jobs, y_d, g2_d)
    torch.cuda.synchronize()
    g2 = g2_d.cpu().numpy().reshape(g.shape)
    assert rel_err(g2, g) < 2e-5


@pytest.mark.parametrize("mu,rb,N,wo,fc", [CASES[0], CASES[2], CASES[5]])
def test_ofdm_symbol_entry_points_equal_the_slot_ones(ctx, mu, rb, N, wo, fc):
    """miphy_ofdm_{de}modulate_symbols (ofdm_symbol_demodulator / _modulator, ofdm_demodulator.h:55-74): one job per OFDM symbol, in any
    order, must give the rows / samples the slot entry points give for the same subframe -- bit-identical, it is the same transform."""
    import torch
    import miphy
    rng = np.random.default_rng(N * 3 + rb)
    nslots = 1 << mu
    nsc = rb * 12
    cfg = miphy.OfdmConfig(mu, rb, N, wo, 0.5, 0.0, fc)
    sizes = [cfg.slot_size(s) for s in range(nslots)]
    sym = [miphy.ofdm_symbol_size(cfg, i) for i in range(14 * nslots)]
    assert [sum(sym[14 * s:14 * s + 14]) for s in range(nslots)] == sizes
    x = ((rng.standard_normal(sum(sizes)) + 1j * rng.standard_normal(sum(sizes))) * 0.7).astype(np.complex64)
    x_d = torch.from_numpy(x).cuda()
    sj = np.zeros(nslots, dtype=miphy.OfdmJob)
    off = 0
    for s in range(nslots):
        sj[s] = (off, s * 14 * nsc, s, 0)
        off += sizes[s]
    g_slot = torch.zeros(nslots * 14 * nsc, dtype=torch.complex64, device="cuda")
    ctx.ofdm_demodulate_slots(cfg, sj, x_d, g_slot)
    order = rng.permutation(14 * nslots)
    starts = np.concatenate([[0], np.cumsum(sym)[:-1]])
    yj = np.zeros(14 * nslots, dtype=miphy.OfdmJob)
    for k, i in enumerate(order):
        yj[k] = (int(starts[i]), int(i) * nsc, int(i), 0)
    g_sym = torch.zeros_like(g_slot)
    ctx.ofdm_demodulate_symbols(cfg, yj, x_d, g_sym)
    torch.cuda.synchronize()
    assert torch.equal(g_slot, g_sym)
    # modulator
    mcfg = miphy.OfdmConfig(mu, rb, N, 0, 0.01, 0.0, fc)
    g = (rng.standard_normal(nslots * 14 * nsc) + 1j * rng.standard_normal(nslots * 14 * nsc)).astype(np.complex64)
    g_d = torch.from_numpy(g).cuda()
    y_slot, y_sym = torch.zeros_like(x_d), torch.zeros_like(x_d)
    ctx.ofdm_modulate_slots(mcfg, sj, g_d, y_slot)
    ctx.ofdm_modulate_symbols(mcfg, yj, g_d, y_sym)
    torch.cuda.synchronize()
    assert torch.equal(y_slot, y_sym)
