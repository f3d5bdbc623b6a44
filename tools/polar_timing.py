"""BASELINE.json configs[3]: PDCCH polar encode + decode (reference-style SSC and SCL-8 with CRC-aided selection), aggregation levels
1-16, batched on one MI355X. Reports codewords/s and the block error rate at the given SNR. usage: python tools/polar_timing.py"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srsran_project_23.5_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, miphy
ctx = miphy.Context()
def timeit(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
n, A = 16384, 40
rng = np.random.default_rng(0)
pay = torch.from_numpy(rng.integers(0, 2, (n, A), dtype=np.uint8)).cuda()
rnti_h = rng.integers(1, 65536, n).astype(np.uint16)
rnti = torch.from_numpy(rnti_h.view(np.int16)).cuda()
for AL in (1, 2, 4, 8, 16):
    E, K = 108 * AL, A + 24
    out = torch.zeros(n * E, dtype=torch.uint8, device="cuda")
    t_enc = timeit(lambda: ctx.pdcch_encode_batch(A, E, n, pay, rnti, out))
    sigma = {1: 0.75, 2: 1.0, 4: 1.4, 8: 2.0, 16: 2.8}[AL]
    y = (1.0 - 2.0 * out.to(torch.float32)) + sigma * torch.randn(n * E, device="cuda")
    llr = torch.clamp(torch.round(y * (2.0 / sigma ** 2) * 4), -120, 120).to(torch.int8)
    code = miphy.PolarCode(K, E, 9, 0)
    msg = torch.zeros(n * K, dtype=torch.uint8, device="cuda")
    t_ssc = timeit(lambda: ctx.polar_decode_batch(code, n, llr, msg))
    ok = torch.zeros(n, dtype=torch.uint8, device="cuda")
    t_scl = timeit(lambda: ctx.polar_decode_list_batch(code, 8, 1, n, llr, rnti, msg, ok))
    torch.cuda.synchronize()
    got = msg.cpu().numpy().reshape(n, K)[:, :A]
    good = (ok.cpu().numpy() != 0) & np.all(got == pay.cpu().numpy(), axis=1)
    false_ok = int(((ok.cpu().numpy() != 0) & ~np.all(got == pay.cpu().numpy(), axis=1)).sum())
    print("AL %2d (K=%d, E=%4d): encode %.3f ms (%.1f M cw/s), SSC decode %.3f ms (%.1f M cw/s), CA-SCL-8 decode %.3f ms (%.1f M cw/s), "
          "SCL-8 BLER %.4f at sigma %.2f, undetected errors %d" % (AL, K, E, t_enc, n / t_enc / 1e3, t_ssc, n / t_ssc / 1e3, t_scl, n / t_scl / 1e3,
                                                                  1 - good.mean(), sigma, false_ok))
