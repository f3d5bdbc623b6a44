"""Cost of the HOST-descriptor entry points for one PDU per call (what the srsRAN adapters issue): miphy_pusch_process_batch on one
273-PRB 256QAM slot, (a) `reps` calls queued back to back with one synchronisation at the end (do the calls wait for the stream?)
and (b) every call followed by a synchronisation (the latency a caller sees). Also the host time spent inside one call.
  python tools/host_api_latency.py [reps]
MIPHY_LIBRARY selects another build of the library (A/B)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srsran_project_23.5_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import miphy
    from test_pusch_proc_gpu import _build_slot, RB_ALL
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    ctx = miphy.Context(0)
    nprb, mod, tbs_bits = 273, 8, 319784
    tb, bg, grid = _build_slot(ctx, miphy, torch, nprb, mod, tbs_bits, 7, 0x4601, 900, 40, 33.0, 0, 10)
    pdus = np.zeros(1, dtype=miphy.PuschPdu)
    p = pdus[0]
    p["numerology"], p["slot_in_frame"], p["rnti"], p["n_id"], p["dmrs_scrambling_id"] = 1, 7, 0x4601, 900, 40
    p["tb_bytes"], p["harq_cb_index"], p["mod"], p["nof_rx_ports"], p["start_symbol"], p["nof_symbols"] = tb.size, 0, mod, 1, 0, 14
    p["bg"], p["rv"], p["new_data"], p["rx_ports"], p["use_early_stop"], p["nof_ldpc_iterations"] = bg, 0, 1, [0, 1, 2, 3], 1, 6
    p["dmrs_symbols_mask"], p["grid_nof_prb"], p["rb_mask"], p["grid_offset"], p["tb_offset"] = 1 << 2, nprb, RB_ALL(nprb), 0, 0
    ncb = miphy.sch_segmentation(tb.size, bg).nof_cbs
    soft = torch.zeros(ncb * miphy.HARQ_CB_STRIDE, dtype=torch.int8, device="cuda")
    msgs = torch.zeros(ncb * miphy.HARQ_MSG_STRIDE, dtype=torch.uint8, device="cuda")
    crc = torch.zeros(ncb, dtype=torch.uint8, device="cuda")
    out = torch.zeros(tb.size, dtype=torch.uint8, device="cuda")
    res = torch.zeros(miphy.PuschResult.itemsize, dtype=torch.uint8, device="cuda")
    sc = torch.zeros(20, dtype=torch.float32, device="cuda")

    def call():
        ctx.pusch_process_batch(pdus, grid, soft, msgs, crc, out, res, sc)

    for _ in range(5):
        call()
    torch.cuda.synchronize()
    assert res.cpu().numpy().view(miphy.PuschResult)[0]["tb_crc_ok"] and np.array_equal(out.cpu().numpy(), tb)
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(reps):
        h0 = time.perf_counter()
        call()
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    queued = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
        torch.cuda.synchronize()
    synced = (time.perf_counter() - t0) / reps
    print("pusch_process_batch, 1 PDU (273 PRB, 256QAM, 38 codeblocks, early stop) per call, host descriptors: queued back to back %.1f us per call "
          "(%.1f us of it inside the call on the host), call + synchronise %.1f us" % (queued * 1e6, host / reps * 1e6, synced * 1e6))


if __name__ == "__main__":
    main()
