"""Achievable HBM rates on this box with library kernels (context for the write-heavy kernels' fractions)."""
import torch
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for mb in (128, 336, 1024):
    n = mb * 1000 * 1000 // 8
    x = torch.empty(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    print("%4d MB: fill %.2f TB/s (write only), copy %.2f TB/s (read+write), sum %.2f TB/s (read only)" % (
        mb, mb * 1e6 / t(lambda: x.fill_(1.0)) / 1e9, 2 * mb * 1e6 / t(lambda: y.copy_(x)) / 1e9, mb * 1e6 / t(lambda: x.sum()) / 1e9))
