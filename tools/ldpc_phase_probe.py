"""Where a packed-decoder codeblock spends its shader cycles (debug build with -DLDPC_PK_PROFILE: s_memtime stamps taken by lane 0 of
wave 0 of every workgroup, summed over the launch). Builds tools/_prof/libmiphy.so when missing (hipcc, here or on the GPU box).
usage: python tools/ldpc_phase_probe.py [layers ...]     (default: the 4-layer headline codeblock and 15 / 46 layers)"""
import ctypes as C, glob, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "srsran_project_23.5_amd")
sys.path.insert(0, PKG)
so = os.path.join(ROOT, "tools", "_prof", "libmiphy.so")
if not os.path.exists(so) or "--rebuild" in sys.argv:
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-DLDPC_PK_PROFILE", "-mllvm", "-enable-post-misched=false", "-shared"]  # the decoder is built without post-RA scheduling (Makefile)
                           + [a for a in sys.argv[1:] if a.startswith("-D")] + [
                           "-o", so] + sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip"))))
if "--build-only" in sys.argv:
    sys.exit(0)
import torch
import miphy
import miphy.binding as B
B.lib_path = so
ctx = miphy.Context()
lib = miphy.lib()
n = 38912
names = ["prologue (descriptor, zero fill, LLR load)", "layer work (wave 0)", "barrier wait after a layer (wave 0)", "final CRC", "hard decision + output",
         "whole codeblock"]
for lay in ([] if "--fused-only" in sys.argv else ([int(a) for a in sys.argv[1:] if a.isdigit() and sys.argv[sys.argv.index(a) - 1] != "--slots"] or [4, 15, 46])):
    bg, Z = 1, 384
    N, K = 66 * Z, 22 * Z
    in_len = min(N, (22 + lay - 2) * Z)
    m = n if lay <= 6 else n // 4
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    llr = (torch.randn(m * in_len, device="cuda", generator=g) * 8 + 10).clamp(-120, 120).to(torch.int8)
    out = torch.zeros(m * (K // 8), dtype=torch.uint8, device="cuda")
    it = torch.zeros(m, dtype=torch.int32, device="cuda")
    d = np.zeros(m, dtype=miphy.LdpcDecDesc)
    for i in range(m):
        d[i] = (bg, miphy.CRC24B, Z, 6, 0, in_len, 1, i * in_len, i * (K // 8))   # flags = 1: CRC once after the last iteration
    dd = torch.from_numpy(d.view(np.uint8)).cuda()
    for rep in range(2):
        buf = (C.c_ulonglong * 8)()
        lib.miphy_debug_ldpc_profile(buf, 1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ctx.ldpc_decode_batch(dd, llr, out, it, limits=(Z, in_len))
        b.record(); torch.cuda.synchronize()
        lib.miphy_debug_ldpc_profile(buf, 0)
    cnt = max(1, buf[7])
    print("BG1 Z=384 %d layers, %d codeblocks, %.3f ms (%.3f us/CB amortised); %d codeblocks stamped" % (lay, m, a.elapsed_time(b), a.elapsed_time(b) * 1e3 / m, cnt))
    for k, nm in enumerate(names):
        print("   %-46s %9.0f cycles per codeblock  (%5.1f %%)" % (nm, buf[k] / cnt, 100.0 * buf[k] / max(1, buf[5])))

# ---- the headline plan (dematch inside the decoder): 1024 transport blocks of 273 PRB / 256QAM / 38 codeblocks, random LLRs
if "--fused" in sys.argv:
    S = int(sys.argv[sys.argv.index("--slots") + 1]) if "--slots" in sys.argv else 1024  # --slots 1: the single slot, latency form of the decoder
    NCB, G, tb_bytes = 38, 273 * 156 * 8, 319784 // 8
    td = np.zeros(S, dtype=miphy.PuschTbDesc)
    for s_ in range(S):
        td[s_] = (1, 0, 8, 1, 1, 0, 6, 0, 273 * 156, tb_bytes, s_ * NCB, s_ * G, s_ * tb_bytes)
    g = torch.Generator(device="cuda"); g.manual_seed(2)
    llr = (torch.randn(S * G, device="cuda", generator=g) * 8 + 10).clamp(-120, 120).to(torch.int8)
    soft = torch.zeros(S * NCB * 66 * 384, dtype=torch.int8, device="cuda")
    msgs = torch.zeros(S * NCB * 1056, dtype=torch.uint8, device="cuda")
    crc = torch.zeros(S * NCB, dtype=torch.uint8, device="cuda")
    tb = torch.zeros(S * tb_bytes, dtype=torch.uint8, device="cuda")
    res = torch.zeros(S * miphy.PuschResult.itemsize, dtype=torch.uint8, device="cuda")
    plan = ctx.pusch_decode_plan(td)
    print("plan info (codeblocks, dematch in decoder, nodes):", plan.info())
    for rep in range(2):
        buf = (C.c_ulonglong * 8)()
        lib.miphy_debug_ldpc_profile(buf, 1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        plan.run(llr, soft, msgs, crc, tb, res)
        b.record(); torch.cuda.synchronize()
        lib.miphy_debug_ldpc_profile(buf, 0)
    cnt = max(1, buf[7])
    print("FUSED plan: %d codeblocks, %.3f ms whole plan; %d codeblocks stamped" % (S * NCB, a.elapsed_time(b), cnt))
    if "--inner" in sys.argv:  # build with -DLDPC_PK_PROFILE2: stamps inside the layer function (thread 0 of every workgroup, the whole run incl. the warm-up pass)
        b2 = (C.c_ulonglong * 8)()
        lib.miphy_debug_ldpc_profile2(b2, 1)
        nm2 = ["scalar edge loads + addresses", "LDS reads issued and returned", "phase 1", "exchange: stores + barrier", "exchange: loads + merge", "scaling + phase 2 + stores", "whole layer function"]
        for k in range(7):
            print("   inner: %-40s %9.0f cycles per codeblock and pass (%5.1f %%)" % (nm2[k], b2[k] / (2.0 * S * NCB), 100.0 * b2[k] / max(1, b2[6])))
    for k, nm in enumerate(names):
        print("   %-46s %9.0f cycles per codeblock  (%5.1f %%)" % (nm, buf[k] / cnt, 100.0 * buf[k] / max(1, buf[5])))
