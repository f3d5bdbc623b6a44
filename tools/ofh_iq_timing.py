"""Times the Open Fronthaul IQ (de)compression kernels (BFP) on the benchmark's grid volume (256 slots x 14 symbols x 273 PRB, 9-bit samples) and
prints the algorithmic bytes per second against the HBM peak. usage (GPU box): python tools/ofh_iq_timing.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "srsran_project_23.5_amd"))
import miphy  # noqa: E402

HBM_PEAK = 8000.0


def main():
    ctx = miphy.Context(0)
    slots, nprb, w = 256, 273, 9
    n = slots * 14
    rec = nprb * (1 + 3 * w)
    jobs = np.zeros(n, dtype=miphy.OfhIqJob)
    for i in range(n):
        jobs[i] = (i * rec, i * nprb * 12, nprb, w, miphy.OFH_COMPRESSION_BFP)
    jd = torch.from_numpy(jobs.view(np.uint8).copy()).cuda()
    x = torch.view_as_complex((torch.randn(n * nprb * 12, 2, device="cuda") * 0.1).clamp(-0.99, 0.99).contiguous())
    p = torch.zeros(n * rec, dtype=torch.uint8, device="cuda")
    y = torch.zeros_like(x)
    algo = n * rec + x.numel() * 8
    for name, fn in (("compress", lambda: ctx.ofh_iq_compress_batch(jd, x, p)), ("decompress", lambda: ctx.ofh_iq_decompress_batch(jd, p, y))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("%-10s %d sections, %.1f MB: %.4f ms  %.0f GB/s (%.0f %% of %.0f GB/s)  %.2f M slots/s" %
              (name, n, algo / 1e6, ms, algo / ms / 1e6, 100 * algo / ms / 1e6 / HBM_PEAK, HBM_PEAK, slots / ms / 1e3))


if __name__ == "__main__":
    main()
