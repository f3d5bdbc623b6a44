"""LDPC decoder throughput against code rate / lifting size (prepared codeblock-level plan = class-sorted launches, 6 iterations, no
early stop): codeblocks/s, information Gbit/s, microseconds per codeblock and the work rate per lane relative to BG1 Z=384 at the same
number of layers (Z <= 64 within 2x of Z = 384). BASELINE.md quotes the reference AVX2 decoder at 450.6 us per BG1
Z=384 codeblock with 25 344 input LLRs. usage: python tools/ldpc_rate_sweep.py [--force 0|1|2] [--n 4096]
  --force 1: one-row-per-lane kernel on every class; --force 2 / 0: packed + wave kernels (class-sorted)"""
import argparse, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srsran_project_23.5_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, miphy
ap = argparse.ArgumentParser(); ap.add_argument("--force", type=int, default=0); ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--small-only", action="store_true")
a = ap.parse_args()
ctx = miphy.Context()
miphy.lib().miphy_debug_force_ldpc_kernel(a.force)
def timeit(f, reps=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
NAMES = {1: "scalar", 2: "packed", 4: "fused", 8: "gmsg", 16: "wave"}
ref = {}  # (bg, layers) -> edge-rows per second of Z = 384
EDGES = {1: [19, 19, 19, 19, 3, 8, 9, 7, 10, 9, 7, 8, 7, 6, 7, 7, 6, 6, 6, 6, 6, 6, 5, 5, 6, 5, 5, 4, 5, 5, 5, 5, 5, 5, 5, 5, 5, 4, 5, 5, 4, 5, 4, 5, 5, 4],
         2: [8, 10, 8, 10, 4, 6, 6, 6, 4, 5, 5, 5, 4, 5, 5, 4, 5, 5, 4, 4, 4, 4, 3, 4, 4, 3, 5, 3, 4, 3, 5, 3, 4, 4, 4, 4, 4, 3, 4, 4, 4, 4]}  # check-row degrees of the base graphs (TS 38.212 tables 5.3.2-2/3)
configs = [(1, 384, (4, 6, 10, 15, 24, 46)), (1, 352, (4, 46)), (1, 208, (4, 46)), (2, 384, (4, 10, 42)), (2, 352, (4, 42)),
           (1, 128, (4, 46)), (1, 96, (4, 46)), (2, 72, (4, 42)), (1, 64, (4, 46)), (2, 64, (4, 42)), (2, 44, (4, 42)), (1, 36, (4, 46)),
           (1, 22, (4, 46)), (1, 15, (4, 46)), (2, 15, (4, 42)), (1, 8, (4, 46)), (2, 3, (4, 42))]
if a.small_only:
    configs = [c for c in configs if c[1] <= 128 or c[1] == 384]
for bg, Z, layers_list in configs:
    n = a.n * 384 // Z  # the same number of soft bits per launch at every lifting size (4096 codeblocks of Z = 384 fill the chip 4 times)
    bgK, nshort = (22, 66) if bg == 1 else (10, 50)
    N, K = nshort * Z, bgK * Z
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    llr = (torch.randn(n * N, device="cuda", generator=g) * 8 + 10).clamp(-120, 120).to(torch.int8)  # mostly-zero codeword at moderate SNR
    out = torch.zeros(n * ((K + 7) // 8), dtype=torch.uint8, device="cuda")
    it = torch.zeros(n, dtype=torch.int32, device="cuda")
    for lay in layers_list:
        in_len = min(N, (bgK + lay - 2) * Z)
        d = np.zeros(n, dtype=miphy.LdpcDecDesc)
        d["bg"], d["crc_poly"], d["Z"], d["max_iter"], d["in_len"] = bg, miphy.CRC_NONE, Z, 6, in_len
        d["llr_offset"] = np.arange(n, dtype=np.uint64) * np.uint64(N)
        d["out_offset"] = np.arange(n, dtype=np.uint64) * np.uint64((K + 7) // 8)
        plan = miphy.LdpcDecodePlan(ctx, d)
        miphy.lib().miphy_debug_ldpc_kernels_used(1)
        ms = timeit(lambda: plan.run(llr, out, it))
        used = int(miphy.lib().miphy_debug_ldpc_kernels_used(1))
        plan.close()
        rows = n * Z * sum(EDGES[bg][:lay]) * 6 / (ms * 1e-3)  # edge-row updates per second
        if Z == 384:
            ref[(bg, lay)] = rows
        rel = ref.get((bg, lay), ref.get((1, lay), rows)) / rows
        print("BG%d Z=%3d %2d layers (in_len %5d, rate %.2f): %7.3f ms per %d CB = %6.3f us/CB, %6.2f M CB/s, %6.1f Gbit/s info, %5.2f T edge-rows/s (Z=384 is %.2fx) [%s]" %
              (bg, Z, lay, in_len, K / (in_len + 2 * Z - 0.0), ms, n, ms * 1e3 / n, n / ms / 1e3, n * K / ms / 1e6, rows / 1e12, rel,
               "+".join(v for k, v in NAMES.items() if used & k)))
