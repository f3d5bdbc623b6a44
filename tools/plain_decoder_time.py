import sys, numpy as np, torch
sys.path.insert(0, "srsran_project_23.5_amd")
import miphy
ctx = miphy.Context()
n, Z, lay = 38912, 384, 4
N, K = 66 * Z, 22 * Z
in_len = (22 + lay - 2) * Z
g = torch.Generator(device="cuda"); g.manual_seed(1)
llr = (torch.randn(n * in_len, device="cuda", generator=g) * 8 + 10).clamp(-120, 120).to(torch.int8)
out = torch.zeros(n * (K // 8), dtype=torch.uint8, device="cuda")
it = torch.zeros(n, dtype=torch.int32, device="cuda")
d = np.zeros(n, dtype=miphy.LdpcDecDesc)
for i in range(n):
    d[i] = (1, miphy.CRC24B, Z, 6, 0, in_len, 1, i * in_len, i * (K // 8))
dd = torch.from_numpy(d.view(np.uint8)).cuda()
for rep in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        ctx.ldpc_decode_batch(dd, llr, out, it, limits=(Z, in_len))
    b.record(); torch.cuda.synchronize()
    print("plain decoder, 38912 CB x 4 layers: %.3f ms per launch" % (a.elapsed_time(b) / 5))
