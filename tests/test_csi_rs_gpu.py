"""GPU: the NZP-CSI-RS generator entry point (miphy_csi_rs_map_batch): bit-exact against grids recorded from the reference generator
(tests/golden/csi_rs.npz: mapping rows 1-8, densities 0.5 / 1 / 3, no CDM / FD-CDM2 / CDM4) as one batch, host and device descriptors."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("on_device", [False, True])
def test_golden_grids_one_batch(ctx, on_device):
    import torch
    import miphy
    g = np.load(os.path.join(GOLD, "csi_rs.npz"))
    n = int(g["n"])
    jobs = np.zeros(n, dtype=miphy.CsiRsJob)
    want, go = [], 0
    for i in range(n):
        slot, scr, amp, start_rb, nof_rb, b, e, st, row, cdm, dens, nports = g["meta_%d" % i]
        j = jobs[i]
        j["slot_in_frame"], j["scrambling_id"], j["amplitude"], j["start_rb"], j["nof_rb"] = int(slot), int(scr), float(amp), int(start_rb), int(nof_rb)
        j["rb_begin"], j["rb_end"], j["rb_stride"], j["grid_nof_prb"], j["mapping_row"], j["cdm"], j["freq_density"] = int(b), int(e), int(st), 80, int(row), int(cdm), int(dens)
        j["nof_ports"] = int(nports)
        j["ports"][:int(nports)] = np.arange(int(nports))
        j["re_mask"][:int(nports)], j["symbol_mask"][:int(nports)] = g["rm_%d" % i], g["sm_%d" % i]
        j["grid_offset"] = go
        want.append(g["grid_%d" % i])
        go += want[-1].size
    gd = torch.zeros(go, dtype=torch.complex64, device="cuda")
    ctx.csi_rs_map_batch(torch.from_numpy(jobs.view(np.uint8).copy()).cuda() if on_device else jobs, gd)
    torch.cuda.synchronize()
    got = gd.cpu().numpy()
    for i, w in enumerate(want):
        o = int(jobs[i]["grid_offset"])
        assert np.array_equal(got[o:o + w.size].view(np.uint32), w.reshape(-1).view(np.uint32)), (i, g["meta_%d" % i])


def test_errors(ctx):
    import torch
    import miphy
    j = np.zeros(1, dtype=miphy.CsiRsJob)
    j["amplitude"], j["nof_rb"], j["rb_end"], j["rb_stride"], j["grid_nof_prb"], j["mapping_row"], j["freq_density"], j["nof_ports"] = 1.0, 24, 24, 1, 52, 2, 2, 1
    j["re_mask"][0][0], j["symbol_mask"][0][0] = 1 << 3, 1 << 5
    g = torch.zeros(14 * 52 * 12, dtype=torch.complex64, device="cuda")
    ctx.csi_rs_map_batch(j, g)
    for field, value, msg in [("nof_ports", 0, "number of ports"), ("nof_ports", 17, "number of ports"), ("cdm", 4, "CDM type"), ("nof_rb", 60, "exceeds the grid"),
                              ("rb_stride", 0, "PRB pattern")]:
        q = j.copy()
        q[field] = value
        with pytest.raises(RuntimeError, match=msg):
            ctx.csi_rs_map_batch(q, g)
