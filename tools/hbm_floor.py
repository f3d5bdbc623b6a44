"""HBM floor for the dematcher's traffic shape: fill N bytes (write-only), copy (read+write), on one MI355X."""
import torch
dev = torch.device("cuda:0")
def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for mb in (64, 246, 1024):
    x = torch.empty(mb * 1024 * 1024, dtype=torch.int8, device=dev)
    y = torch.empty_like(x)
    ms = t(lambda: x.zero_())
    print(f"fill  {mb:5d} MiB: {ms:.4f} ms -> {mb*1.048576/ms:.1f} GB/s written")
    ms = t(lambda: y.copy_(x))
    print(f"copy  {mb:5d} MiB: {ms:.4f} ms -> {2*mb*1.048576/ms:.1f} GB/s read+written")
