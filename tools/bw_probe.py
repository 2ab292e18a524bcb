"""Achievable HBM streaming rates on this GPU with stock torch kernels (fill, copy),
as a yardstick for the memory-bound kernels."""
import torch

def t(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

for mb in (64, 177, 354, 1024, 4096):
    n = mb * (1 << 20) // 2
    a = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    b = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    tf = t(lambda: a.fill_(1.0))
    tc = t(lambda: b.copy_(a))
    tr = t(lambda: a.view(torch.int32).sum())
    print(f"{mb:5d} MB  fill {mb / 1024 / tf / 1e3 * 1.0737:6.2f} TB/s   copy(r+w) {2 * mb / 1024 / tc / 1e3 * 1.0737:6.2f} TB/s   "
          f"read-sum {mb / 1024 / tr / 1e3 * 1.0737:6.2f} TB/s", flush=True)
