"""Time the first-layer kernel (3 -> 32 channels, 416x416, batch 32) back to back
through the C ABI, beside a torch fill of the same output tensor (the write-only
floor of the layer)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from yolo_amd import _lib
from yolo_amd._lib import ConvDesc, check, lib

B, H = 32, 416
x = torch.rand((B, 3, H, H), device="cuda")
wk = (torch.randn((27, 32), device="cuda") * 0.2).contiguous()
bk = torch.randn(32, device="cuda")
out = torch.empty((B, H, H, 32), dtype=torch.bfloat16, device="cuda")
d = (ConvDesc * 1)()
d[0].src0, d[0].weight, d[0].bias, d[0].dst = x.data_ptr(), wk.data_ptr(), bk.data_ptr(), out.data_ptr()
d[0].c0, d[0].c1, d[0].up0, d[0].B, d[0].H, d[0].W = 3, 0, 0, B, H, H
d[0].cout, d[0].cout_pad, d[0].ksize, d[0].stride, d[0].k_pad = 32, 32, 3, 1, 27
d[0].leaky, d[0].dst_fp32, d[0].first_layer = 1, 0, 1
st = _lib.current_stream_ptr(x.device)


def many(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


t = many(lambda: check(lib.yolo_amd_conv_forward(d, 1, None, st), "first conv"))
print(f"first layer {t:7.1f} us  ({out.numel() * 2 / t / 1e6:.2f} TB/s of output)")
t = many(lambda: out.fill_(1.0))
print(f"torch fill  {t:7.1f} us  ({out.numel() * 2 / t / 1e6:.2f} TB/s)")
