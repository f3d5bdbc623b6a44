"""GPU: the PDCCH processor entry point (miphy_pdcch_process_batch): DCI payloads to grid REs, bit-exact against grids recorded
from the reference processor (tests/golden/pdcch_proc.npz, all three CCE-to-REG mapping types) and against the oracle on a batch of
random PDUs sharing one grid buffer."""
import os

import numpy as np
import pytest

import oracle_lib as O
from test_pdsch_mod_gpu import _words

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _pdu(miphy, slot, rnti, nd, nr, ndm, ref, xdb, ddb, A, AL, start, dur, rb, pay_off, grid_off, port=0):
    p = np.zeros(1, dtype=miphy.PdcchPdu)[0]
    p["slot_in_frame"], p["rnti"], p["n_id_pdcch_data"], p["n_rnti"], p["n_id_pdcch_dmrs"], p["reference_point_k_rb"] = slot, rnti, nd, nr, ndm, ref
    p["data_power_offset_dB"], p["dmrs_power_offset_dB"], p["payload_size"], p["aggregation_level"] = xdb, ddb, A, AL
    p["start_symbol"], p["duration"], p["port"], p["grid_nof_prb"], p["rb_mask"] = start, dur, port, rb.size, _words(rb)
    p["payload_offset"], p["grid_offset"] = pay_off, grid_off
    return p


def test_golden_grids_one_batch(ctx):
    import torch
    import miphy
    g = np.load(os.path.join(GOLD, "pdcch_proc.npz"))
    n = int(g["n"])
    pdus, pays, want, po, go = [], [], [], 0, 0
    for i in range(n):
        slot, rnti, nd, nr, ndm, ref, xdb, ddb, AL, start, dur, mapping = g["meta_%d" % i]
        pay, rb, grid = g["pay_%d" % i], g["rb_%d" % i], g["grid_%d" % i]
        pdus.append(_pdu(miphy, int(slot), int(rnti), int(nd), int(nr), int(ndm), int(ref), float(xdb), float(ddb), pay.size, int(AL), int(start), int(dur), rb, po, go))
        pays.append(pay)
        want.append(grid)
        po += pay.size
        go += grid.size
    gd = torch.zeros(go, dtype=torch.complex64, device="cuda")
    ctx.pdcch_process_batch(np.array(pdus, dtype=miphy.PdcchPdu), torch.from_numpy(np.concatenate(pays)).cuda(), gd)
    torch.cuda.synchronize()
    got = gd.cpu().numpy()
    for i, (p, w) in enumerate(zip(pdus, want)):
        o = int(p["grid_offset"])
        assert np.array_equal(got[o:o + w.size].view(np.uint32), w.reshape(-1).view(np.uint32)), i


def test_random_pdus_match_oracle(ctx):
    import torch
    import miphy
    rng = np.random.default_rng(404)
    nprb = 106
    pdus, pays, po = [], [], 0
    want = np.zeros((2, 14, nprb * 12), dtype=np.complex64)
    used = np.zeros((2, 3, nprb), bool)
    for _ in range(400):  # place as many non-overlapping candidates as fit (bounded: the grid fills up)
        if len(pdus) == 14:
            break
        AL, dur, port = int(rng.choice([1, 2, 4, 8, 16])), int(rng.integers(1, 4)), int(rng.integers(0, 2))
        if (6 * AL) % dur:
            continue
        n_rb = 6 * AL // dur
        if n_rb > nprb:
            continue
        cand = np.nonzero(~used[port, :dur].any(axis=0))[0]
        if cand.size < n_rb:
            continue
        rb = np.zeros(nprb, np.uint8)
        rb[rng.choice(cand, n_rb, replace=False)] = 1
        used[port, :dur] |= rb.astype(bool)
        A = int(rng.integers(12, min(129, 108 * AL - 24)))
        pay = rng.integers(0, 2, A, dtype=np.uint8)
        slot, rnti, nd, nr, ndm = int(rng.integers(0, 20)), int(rng.integers(1, 65536)), int(rng.integers(0, 65536)), int(rng.integers(0, 65536)), int(rng.integers(0, 65536))
        ref = int(rng.integers(0, int(np.nonzero(rb)[0][0]) + 1))
        xdb, ddb = float(rng.choice([0.0, -3.0, 1.5])), float(rng.choice([0.0, 3.0]))
        assert O.o_pdcch_process(slot, rnti, nd, nr, ndm, ref, xdb, ddb, pay, AL, 0, dur, rb, want[port]) == 54 * AL
        pdus.append(_pdu(miphy, slot, rnti, nd, nr, ndm, ref, xdb, ddb, A, AL, 0, dur, rb, po, 0, port))
        pays.append(pay)
        po += A
    assert len(pdus) >= 6 and len({int(p["aggregation_level"]) for p in pdus}) >= 3
    gd = torch.zeros(want.size, dtype=torch.complex64, device="cuda")
    ctx.pdcch_process_batch(np.array(pdus, dtype=miphy.PdcchPdu), torch.from_numpy(np.concatenate(pays)).cuda(), gd)
    torch.cuda.synchronize()
    assert np.array_equal(gd.cpu().numpy().view(np.uint32), want.reshape(-1).view(np.uint32))


def test_errors(ctx):
    import torch
    import miphy
    rb = np.zeros(24, np.uint8)
    rb[:6] = 1
    ok = _pdu(miphy, 0, 1, 2, 3, 4, 0, 0.0, 0.0, 40, 1, 0, 1, rb, 0, 0)
    pl = torch.zeros(256, dtype=torch.uint8, device="cuda")
    g = torch.zeros(14 * 24 * 12, dtype=torch.complex64, device="cuda")
    ctx.pdcch_process_batch(np.array([ok], dtype=miphy.PdcchPdu), pl, g)
    for field, value, msg in [("aggregation_level", 3, "aggregation level"), ("duration", 4, "CORESET duration"), ("payload_size", 11, "payload size"),
                              ("payload_size", 129, "payload size"), ("aggregation_level", 2, "do not match aggregation level"), ("rnti", 70000, "identifier")]:
        q = np.array([ok], dtype=miphy.PdcchPdu)
        q[0][field] = value
        with pytest.raises(RuntimeError, match=msg):
            ctx.pdcch_process_batch(q, pl, g)
