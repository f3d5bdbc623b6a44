"""Times the transport-block level entry points on the bench workload: miphy_pusch_decode_batch (256 TBs of 38 codeblocks) and
miphy_pdsch_encode_batch, with rocprof-independent HIP events. usage: python tools/tb_level_timing.py"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srsran_project_23.5_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, miphy
ctx = miphy.Context()
S, tbs_bits, nsym, mod = 256, 319784, 273 * 156, 8
G, C, tb_bytes = nsym * mod, 38, 319784 // 8
rng = np.random.default_rng(0)
tb = rng.integers(0, 256, 4 * tb_bytes, dtype=np.uint8)
td = np.zeros(S, dtype=miphy.PdschTbDesc)
for s in range(S):
    td[s] = (1, 0, mod, 1, 0, nsym, tb_bytes, (s % 4) * tb_bytes, s * G)
tb_d = torch.from_numpy(tb).cuda()
cw_d = torch.zeros(S * G, dtype=torch.uint8, device="cuda")
def timeit(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ms = timeit(lambda: ctx.pdsch_encode_batch(td, tb_d, cw_d))
print("pdsch_encode_batch: %.3f ms per %d TBs -> %.1f Gbit/s information bits" % (ms, S, S * tbs_bits / ms / 1e6))
llr = ((1.0 - 2.0 * cw_d.to(torch.float32)) * 40).to(torch.int8)
pd = np.zeros(S, dtype=miphy.PuschTbDesc)
for s in range(S):
    pd[s] = (1, 0, mod, 1, 1, 0, 6, 0, nsym, tb_bytes, s * C, s * G, s * tb_bytes)
soft = torch.zeros(S * C * miphy.HARQ_CB_STRIDE, dtype=torch.int8, device="cuda")
msgs = torch.zeros(S * C * 1056, dtype=torch.uint8, device="cuda")
crc = torch.zeros(S * C, dtype=torch.uint8, device="cuda")
out = torch.zeros(S * tb_bytes, dtype=torch.uint8, device="cuda")
res = torch.zeros(S * miphy.PuschResult.itemsize, dtype=torch.uint8, device="cuda")
ms = timeit(lambda: ctx.pusch_decode_batch(pd, llr, soft, msgs, crc, out, res))
r = res.cpu().numpy().view(miphy.PuschResult)
ok = int((r["tb_crc_ok"] != 0).sum())
same = bool(np.array_equal(out.cpu().numpy().reshape(S, tb_bytes)[5], tb[tb_bytes:2 * tb_bytes]))
print("pusch_decode_batch: %.3f ms per %d TBs -> %.1f Gbit/s information bits; %d/%d TB CRC ok, TB bytes equal %s" % (ms, S, S * tbs_bits / ms / 1e6, ok, S, same))
