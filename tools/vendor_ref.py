"""Calibration, not product: what the vendor libraries reach on this GPU for the shapes of the 3x3 layers --
MIOpen through torch.nn.functional.conv2d (bf16, channels_last, bias and LeakyReLU as separate kernels NOT timed)
and hipBLASLt/rocBLAS through torch.matmul on the im2col-equivalent GEMM (M = pixels, K = 9 Cin, N = Cout; no
gather, the best case) -- next to this library's kernel for the same layer.
    python tools/vendor_ref.py > gpurun_out/vendor_ref.txt"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import conv_bench as cb  # noqa: E402
from yolo_amd import _lib  # noqa: E402
from yolo_amd._lib import check, lib  # noqa: E402


def timed(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3       # us


def main():
    dev = torch.device("cuda", 0)
    B = 32
    zero = torch.zeros(1024, dtype=torch.uint8, device=dev)
    stream = _lib.current_stream_ptr(dev)
    print(f"{'layer':22s} {'this library':>14s} {'MIOpen conv2d':>15s} {'GEMM (matmul)':>15s}   us and TFLOP/s, batch {B}")
    for name in ("3x3  64->128 @104", "3x3 128->256 @52", "3x3 256->512 @26", "3x3 512->1024 @13"):
        spec = [s for s in cb.LAYERS if s[0] == name][0]
        d, keep, flops, byts = cb.make_desc(B, *spec[1:], dev)
        ours = timed(lambda: check(lib.yolo_amd_conv_forward(d, 1, zero.data_ptr(), stream), "conv"))
        H, cin, cout = spec[1], spec[2], spec[3]
        x = torch.randn(B, cin, H, H, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, 3, 3, device=dev, dtype=torch.bfloat16) * 0.05).contiguous(memory_format=torch.channels_last)
        mi = timed(lambda: F.conv2d(x, w, None, 1, 1))
        a = torch.randn(B * H * H, 9 * cin, device=dev, dtype=torch.bfloat16)
        bm = torch.randn(9 * cin, cout, device=dev, dtype=torch.bfloat16) * 0.05
        mm = timed(lambda: torch.matmul(a, bm))
        tf = lambda us: flops / (us * 1e-6) / 1e12
        print(f"{name:22s} {ours:7.1f} {tf(ours):6.0f} {mi:8.1f} {tf(mi):6.0f} {mm:8.1f} {tf(mm):6.0f}")
    # the 1x1 layers are plain products already: the GEMM library on their shapes (no bias, no activation)
    print(f"{'layer':22s} {'this library':>14s} {'GEMM (matmul)':>15s}")
    for name in ("1x1  64->32  @208", "1x1 128->64  @104", "1x1 256->128 @52", "1x1 512->256 @26", "1x1 1024->512 @13"):
        spec = [s for s in cb.LAYERS if s[0] == name]
        if not spec:
            continue
        spec = spec[0]
        d, keep, flops, byts = cb.make_desc(B, *spec[1:], dev)
        ours = timed(lambda: check(lib.yolo_amd_conv_forward(d, 1, zero.data_ptr(), stream), "conv"))
        H, cin, cout = spec[1], spec[2], spec[3]
        a = torch.randn(B * H * H, cin, device=dev, dtype=torch.bfloat16)
        bm = torch.randn(cin, cout, device=dev, dtype=torch.bfloat16) * 0.05
        mm = timed(lambda: torch.matmul(a, bm))
        tf = lambda us: flops / (us * 1e-6) / 1e12
        print(f"{name:22s} {ours:7.1f} {tf(ours):6.0f} {mm:8.1f} {tf(mm):6.0f}")
    # what the GEMM library sustains on this GPU when the shape is large and square
    for n in (4096, 8192):
        a = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
        b2 = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
        us = timed(lambda: torch.matmul(a, b2), reps=10, warm=3)
        print(f"matmul {n}^3 bf16: {us:.0f} us, {2.0 * n ** 3 / (us * 1e-6) / 1e12:.0f} TFLOP/s")


if __name__ == "__main__":
    main()
