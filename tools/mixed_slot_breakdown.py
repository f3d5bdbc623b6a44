"""Decode time of the mixed slot of bench_legs.MIXED_PDUS per PDU type (1024 transport blocks of ONE type per plan) and for all eight together: shows which launch
classes cost what, and what running the classes one after another loses against their sum. usage (GPU box): python tools/mixed_slot_breakdown.py"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srsran_project_23.5_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch, miphy, bench_legs as BL
ctx = miphy.Context(0); dev = torch.device("cuda", 0)
S = 1024
tot = 0.0
def run(pdus, label):
    segs = [miphy.sch_segmentation(t // 8, bg) for (_, _, _, t, bg, _) in pdus]
    G = [n * 156 * m for (_, n, m, _, _, _) in pdus]; C = [sg.nof_cbs for sg in segs]; tbb = [t // 8 for (_, _, _, t, _, _) in pdus]
    Gs, Cs, Ts = sum(G), sum(C), sum((b + 15) // 16 * 16 for b in tbb)
    td = np.zeros(S * len(pdus), dtype=miphy.PuschTbDesc)
    for s in range(S):
        go = co = to = 0
        for u, (_, n, m, t, bg, _) in enumerate(pdus):
            td[s * len(pdus) + u] = (bg, 0, m, 1, 1, 0, 6, 0, n * 156, tbb[u], s * Cs + co, s * Gs + go, s * Ts + to)
            go += G[u]; co += C[u]; to += (tbb[u] + 15) // 16 * 16
    g = torch.Generator(device=dev); g.manual_seed(1)
    llr = (torch.randn(S * Gs, device=dev, generator=g) * 8 + 10).clamp(-120, 120).to(torch.int8)
    soft = torch.zeros(S * Cs * miphy.HARQ_CB_STRIDE, dtype=torch.int8, device=dev); msgs = torch.zeros(S * Cs * miphy.HARQ_MSG_STRIDE, dtype=torch.uint8, device=dev)
    crc = torch.zeros(S * Cs, dtype=torch.uint8, device=dev); tb = torch.zeros(S * Ts, dtype=torch.uint8, device=dev)
    res = torch.zeros(S * len(pdus) * miphy.PuschResult.itemsize, dtype=torch.uint8, device=dev)
    plan = ctx.pusch_decode_plan(td); plan.enable_timing(16)
    BL.ev_ms(torch, lambda: plan.run(llr, soft, msgs, crc, tb, res), 5)
    tm = plan.read_timing()
    print("%-34s CB/slot %2d Z %s launches %d: dematch %.3f decode %.3f assemble %.3f ms" % (label, Cs, sorted({sg.Z for sg in segs}), plan.nof_launches(), tm["rate_dematch"], tm["ldpc_decode"], tm["tb_assemble"]))
    plan.close()
    return tm["ldpc_decode"] + tm["rate_dematch"]
for p in BL.MIXED_PDUS:
    tot += run([p], "PRB %3d Qm %d R %4d TBS %6d" % (p[1], p[2], p[5], p[3]))
print("sum of the separate types: %.3f ms" % tot)
for ns in (1, 2, 4, 1, 4):
    miphy.lib().miphy_debug_set_ldpc_class_streams(ns)
    run(BL.MIXED_PDUS, "all eight per slot, %d streams" % ns)
