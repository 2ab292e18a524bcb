"""Times yolo_amd_preprocess on synthetic frames already resident in HBM.
    python tools/pre_bench.py [--batch 32] [--w 1920 --h 1080] [--side 416]
Prints frames/s, microseconds per call and the HBM rate of the algorithmic bytes
(source pixels touched once + canvas written and read + fp32 input written)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd.harness import IMAGENET_MEAN, IMAGENET_STD, preprocess_frames  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--side", type=int, default=416)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(0)
    frames = [torch.randint(0, 256, (a.h, a.w, 3), dtype=torch.uint8, device="cuda", generator=g)
              for _ in range(a.batch)]
    for _ in range(5):
        preprocess_frames(frames, a.side, mean=IMAGENET_MEAN, std=IMAGENET_STD)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        preprocess_frames(frames, a.side, mean=IMAGENET_MEAN, std=IMAGENET_STD)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    scale = min(a.side / a.w, a.side / a.h)
    nw, nh = int(a.w * scale), int(a.h * scale)
    # rows of the source a 4-tap column touches: all of them when shrinking by < 4x
    src_rows = min(a.h, 4 * nh)
    src_bytes = src_rows * min(a.w, 4 * nw) * 3
    alg = a.batch * (src_bytes + 2 * a.side * a.side * 3 + a.side * a.side * 12)
    # the library call alone, descriptors prepared once
    import ctypes
    from yolo_amd import _lib
    descs = (_lib.ImageDesc * a.batch)()
    for i, t in enumerate(frames):
        descs[i].data, descs[i].width, descs[i].height, descs[i].pitch_bytes = t.data_ptr(), a.w, a.h, t.stride(0)
    out = torch.empty((a.batch, 3, a.side, a.side), device="cuda")
    n = _lib.lib.yolo_amd_preprocess_workspace_bytes(a.batch, a.side)
    ws = torch.empty(n, dtype=torch.uint8, device="cuda")
    m, sd = (ctypes.c_float * 3)(*IMAGENET_MEAN), (ctypes.c_float * 3)(*IMAGENET_STD)
    st = _lib.current_stream_ptr(out.device)
    call = lambda: _lib.lib.yolo_amd_preprocess(descs, a.batch, a.side, 1, m, sd, out.data_ptr(), None,
                                                ws.data_ptr(), n, st)
    for _ in range(5):
        assert call() == 0
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    us_lib = e0.elapsed_time(e1) * 1e3 / a.iters
    print(f"library call alone: {us_lib:.1f} us, {alg / us_lib / 1e3:.1f} GB/s")
    print(f"{a.batch} frames {a.w}x{a.h} -> {a.side}: {us:.1f} us/call, "
          f"{a.batch / us * 1e6:.0f} frames/s, {alg / us / 1e3:.1f} GB/s of {alg / 1e6:.1f} MB")


if __name__ == "__main__":
    main()
