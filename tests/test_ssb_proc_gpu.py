"""GPU: the SS/PBCH block processor entry point (miphy_ssb_process_batch): bit-exact against grids recorded from the reference
processor (tests/golden/ssb_proc.npz) and against the oracle on a batch of random blocks written to two ports."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _pdu(miphy, N_id, ssb_idx, L_max, hrf, sfn, kssb, pay, k0, l0, beta, nprb, ports, grid_off):
    p = np.zeros(1, dtype=miphy.SsbPdu)[0]
    p["msg"] = (N_id, ssb_idx, L_max, hrf, sfn, kssb, pay)
    p["ssb_first_subcarrier"], p["ssb_first_symbol"], p["beta_pss_dB"], p["grid_nof_prb"], p["nof_ports"] = k0, l0, beta, nprb, len(ports)
    p["ports"][:len(ports)] = ports
    p["grid_offset"] = grid_off
    return p


def test_golden_grids_one_batch(ctx):
    import torch
    import miphy
    g = np.load(os.path.join(GOLD, "ssb_proc.npz"))
    n = int(g["n"])
    pdus, want, go = [], [], 0
    for i in range(n):
        N_id, ssb_idx, L_max, hrf, sfn, kssb, k0, l0, beta, case = g["meta_%d" % i]
        pdus.append(_pdu(miphy, int(N_id), int(ssb_idx), int(L_max), int(hrf), int(sfn), int(kssb), g["pay_%d" % i], int(k0), int(l0), float(beta), 106, [0], go))
        want.append(g["grid_%d" % i])
        go += want[-1].size
    gd = torch.zeros(go, dtype=torch.complex64, device="cuda")
    ctx.ssb_process_batch(np.array(pdus, dtype=miphy.SsbPdu), gd)
    torch.cuda.synchronize()
    got = gd.cpu().numpy()
    for i, (p, w) in enumerate(zip(pdus, want)):
        o = int(p["grid_offset"])
        assert np.array_equal(got[o:o + w.size].view(np.uint32), w.reshape(-1).view(np.uint32)), i


def test_random_blocks_two_ports_match_oracle(ctx):
    import torch
    import miphy
    rng = np.random.default_rng(505)
    nprb, n = 52, 12
    pdus, want = [], []
    for i in range(n):
        N_id, L_max = int(rng.integers(0, 1008)), int(rng.choice([4, 8, 64]))
        ssb_idx, hrf, sfn, kssb = int(rng.integers(0, L_max)), int(rng.integers(0, 2)), int(rng.integers(0, 1024)), int(rng.integers(0, 24))
        k0, l0, beta = int(rng.integers(0, nprb * 12 - 240 + 1)), int(rng.integers(0, 11)), float(rng.choice([0.0, 3.0, -3.0]))
        pay = rng.integers(0, 2, 32, dtype=np.uint8)
        one = np.zeros((14, nprb * 12), dtype=np.complex64)
        assert O.o_ssb_process(N_id, ssb_idx, L_max, hrf, sfn, kssb, pay, k0, l0, beta, nprb, one) == 0
        g = np.zeros((3, 14, nprb * 12), dtype=np.complex64)
        g[0], g[2] = one, one
        want.append(g)
        pdus.append(_pdu(miphy, N_id, ssb_idx, L_max, hrf, sfn, kssb, pay, k0, l0, beta, nprb, [0, 2], i * g.size))
    gd = torch.zeros(n * want[0].size, dtype=torch.complex64, device="cuda")
    ctx.ssb_process_batch(np.array(pdus, dtype=miphy.SsbPdu), gd)
    torch.cuda.synchronize()
    assert np.array_equal(gd.cpu().numpy().view(np.uint32), np.concatenate([w.reshape(-1) for w in want]).view(np.uint32))


def test_errors(ctx):
    import torch
    import miphy
    g = torch.zeros(14 * 52 * 12, dtype=torch.complex64, device="cuda")
    ok = _pdu(miphy, 5, 1, 4, 0, 10, 3, np.zeros(32, np.uint8), 0, 2, 0.0, 52, [0], 0)
    ctx.ssb_process_batch(np.array([ok], dtype=miphy.SsbPdu), g)
    for mutate, msg in [(lambda p: p["msg"].__setitem__("N_id", 1008), "cell identity"), (lambda p: p["msg"].__setitem__("L_max", 5), "L_max"),
                        (lambda p: p.__setitem__("nof_ports", 0), "number of ports"), (lambda p: p.__setitem__("ssb_first_subcarrier", 52 * 12 - 239), "fit the grid"),
                        (lambda p: p.__setitem__("ssb_first_symbol", 11), "fit the slot")]:
        q = np.array([ok], dtype=miphy.SsbPdu)
        mutate(q[0])
        with pytest.raises(RuntimeError, match=msg):
            ctx.ssb_process_batch(q, g)
