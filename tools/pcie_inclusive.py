"""PCIe-inclusive rate of the bench step: time-domain samples of S slots from pinned host memory to the GPU, the step, transport
blocks back. Uses the measured step time of bench.py (argument, ms). usage: python tools/pcie_inclusive.py 1.41"""
import sys, torch
S, slot_samples, tb_bytes = 256, 61440, 319784 // 8
step_ms = float(sys.argv[1]) if len(sys.argv) > 1 else 1.41
h_in = torch.empty(S * slot_samples, dtype=torch.complex64).pin_memory()
h_out = torch.empty(S * tb_bytes, dtype=torch.uint8).pin_memory()
d_in = torch.empty_like(h_in, device="cuda")
d_out = torch.empty(S * tb_bytes, dtype=torch.uint8, device="cuda")
def t(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
h2d = t(lambda: d_in.copy_(h_in, non_blocking=True))
d2h = t(lambda: h_out.copy_(d_out, non_blocking=True))
bits = S * 319784
print("H2D %.1f MB in %.3f ms (%.1f GB/s), D2H %.1f MB in %.3f ms (%.1f GB/s)" % (h_in.numel() * 8 / 1e6, h2d, h_in.numel() * 8 / h2d / 1e6,
                                                                              h_out.numel() / 1e6, d2h, h_out.numel() / d2h / 1e6))
print("step %.3f ms -> serial H2D + step + D2H: %.2f Gbit/s; copies overlapped with the previous / next step (bound by the slowest stage): %.2f Gbit/s" %
      (step_ms, bits / (h2d + step_ms + d2h) / 1e6, bits / max(h2d, step_ms, d2h) / 1e6))
