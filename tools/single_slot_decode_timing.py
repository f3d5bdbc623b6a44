"""Decoder time of ONE slot (38 codeblocks BG1 Z=384, 4 layers, 6 iterations, dematch in the decoder) through the transport-block plan:
latency form of the packed kernel (automatic choice for launches of at most one codeblock per CU) against the throughput form.
usage (GPU box): python tools/single_slot_decode_timing.py"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srsran_project_23.5_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch, miphy, bench_legs as BL
ctx = miphy.Context(0); dev = torch.device("cuda", 0)
for name, (bg, mod, nprb, tbs) in {"273 PRB 256QAM R=948 (38 CB, 4 layers)": (1, 8, 273, 319784), "273 PRB 16QAM R=658 (13 CB, 15 layers)": (1, 4, 273, 108552),
                                    "106 PRB 64QAM R=873 (10 CB)": (1, 6, 106, 83976)}.items():
    G, tb_bytes = nprb * 156 * mod, tbs // 8
    sg = miphy.sch_segmentation(tb_bytes, bg); C = sg.nof_cbs
    td = np.zeros(1, dtype=miphy.PuschTbDesc); td[0] = (bg, 0, mod, 1, 1, 0, 6, 0, nprb * 156, tb_bytes, 0, 0, 0)
    g = torch.Generator(device=dev); g.manual_seed(1)
    llr = (torch.randn(G, device=dev, generator=g) * 8 + 10).clamp(-120, 120).to(torch.int8)
    soft = torch.zeros(C * miphy.HARQ_CB_STRIDE, dtype=torch.int8, device=dev); msgs = torch.zeros(C * miphy.HARQ_MSG_STRIDE, dtype=torch.uint8, device=dev)
    crc = torch.zeros(C, dtype=torch.uint8, device=dev); tb = torch.zeros(tb_bytes, dtype=torch.uint8, device=dev); res = torch.zeros(miphy.PuschResult.itemsize, dtype=torch.uint8, device=dev)
    for force, label in ((4, "throughput form"), (0, "automatic (latency form)")):
        miphy.lib().miphy_debug_force_ldpc_kernel(force)
        plan = ctx.pusch_decode_plan(td); plan.enable_timing(64)
        miphy.lib().miphy_debug_ldpc_kernels_used(1)
        ms = BL.ev_ms(torch, lambda: plan.run(llr, soft, msgs, crc, tb, res), 20)
        used = int(miphy.lib().miphy_debug_ldpc_kernels_used(1)); tm = plan.read_timing(); plan.close()
        print("%-42s %-26s kernels %2d: plan %.1f us, decode %.1f us, assemble %.1f us" % (name, label, used, ms * 1e3, tm["ldpc_decode"] * 1e3, tm["tb_assemble"] * 1e3))
miphy.lib().miphy_debug_force_ldpc_kernel(0)
