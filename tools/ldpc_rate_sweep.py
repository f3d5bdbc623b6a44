"""LDPC decoder throughput against code rate / lifting size (codeblock-level API, device descriptors with limits, 6 iterations, no
early stop): codeblocks/s, information Gbit/s and microseconds per codeblock. BASELINE.md quotes the reference AVX2 decoder at
450.6 us per BG1 Z=384 codeblock with 25 344 input LLRs. usage: python tools/ldpc_rate_sweep.py"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srsran_project_23.5_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, miphy
ctx = miphy.Context()
def timeit(f, reps=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
n = 4096
for bg, Z, layers_list in ((1, 384, (4, 6, 10, 15, 24, 46)), (1, 352, (4, 46)), (1, 208, (4, 46)), (2, 384, (4, 10, 42)), (2, 352, (4, 42)),
                           (1, 96, (4, 46)), (2, 44, (4, 42)), (1, 15, (4, 46))):
    bgK, nshort = (22, 66) if bg == 1 else (10, 50)
    N, K = nshort * Z, bgK * Z
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    llr = (torch.randn(n * N, device="cuda", generator=g) * 8 + 10).clamp(-120, 120).to(torch.int8)  # mostly-zero codeword at moderate SNR
    out = torch.zeros(n * ((K + 7) // 8), dtype=torch.uint8, device="cuda")
    it = torch.zeros(n, dtype=torch.int32, device="cuda")
    for lay in layers_list:
        in_len = min(N, (bgK + lay - 2) * Z)
        d = np.zeros(n, dtype=miphy.LdpcDecDesc)
        for i in range(n):
            d[i] = (bg, miphy.CRC_NONE, Z, 6, 0, in_len, 0, i * N, i * ((K + 7) // 8))
        dd = torch.from_numpy(d.view(np.uint8)).cuda()
        ms = timeit(lambda: ctx.ldpc_decode_batch(dd, llr, out, it, limits=(Z, in_len)))
        print("BG%d Z=%3d %2d layers (in_len %5d, rate %.2f): %7.3f ms per %d CB = %6.2f us/CB, %6.2f M CB/s, %6.1f Gbit/s info" %
              (bg, Z, lay, in_len, K / (in_len + 2 * Z - 0.0), ms, n, ms * 1e3 / n, n / ms / 1e3, n * K / ms / 1e6))
