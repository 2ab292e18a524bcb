"""Time single fused conv layers of the YOLOv3-416 / batch-32 workload through
the C ABI, for kernel variants selected with the YOLO_AMD_CONV_* development
knobs (all variants run interleaved in one process, as A/B timing should).

    python tools/conv_bench.py [--reps 20] [--batch 32] [--variants "ns=2;ns=4;ns=4,bk=32"]
"""
from __future__ import annotations

import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from yolo_amd import _lib
from yolo_amd._lib import ConvDesc, check, lib
from yolo_amd.engine import pack_conv_weights

# name, H(in), cin, cout, k, stride, residual
LAYERS = [
    ("3x3  32->64  @208", 208, 32, 64, 3, 1, True),
    ("3x3s2 32->64 @416", 416, 32, 64, 3, 2, False),
    ("3x3  64->128 @104", 104, 64, 128, 3, 1, True),
    ("3x3 128->256 @52", 52, 128, 256, 3, 1, True),
    ("3x3 256->512 @26", 26, 256, 512, 3, 1, True),
    ("3x3 512->1024 @13", 13, 512, 1024, 3, 1, True),
    ("608 3x3 128->256 @76", 76, 128, 256, 3, 1, True),
    ("608 3x3 256->512 @38", 38, 256, 512, 3, 1, True),
    ("608 3x3 512->1024 @19", 19, 512, 1024, 3, 1, True),
    ("3x3s2 64->128 @208", 208, 64, 128, 3, 2, False),
    ("3x3s2 128->256 @104", 104, 128, 256, 3, 2, False),
    ("3x3s2 256->512 @52", 52, 256, 512, 3, 2, False),
    ("3x3s2 512->1024 @26", 26, 512, 1024, 3, 2, False),
    ("1x1  64->32  @208", 208, 64, 32, 1, 1, False),
    ("1x1 128->64  @104", 104, 128, 64, 1, 1, False),
    ("1x1 256->128 @52", 52, 256, 128, 1, 1, False),
    ("1x1 512->256 @26", 26, 512, 256, 1, 1, False),
    ("1x1 1024->512 @13", 13, 1024, 512, 1, 1, False),
    ("1x1 256->255 @52 fp32", 52, 256, 255, 1, 1, False),
    ("1x1 512->255 @26 fp32", 26, 512, 255, 1, 1, False),
    ("1x1 1024->255 @13 fp32", 13, 1024, 255, 1, 1, False),
    # almost no K loop: what a launch costs in set-up, epilogue and tail
    ("ovh 1x1 32->512 @26 res", 26, 32, 512, 1, 1, True),
    ("ovh 1x1 32->512 @26", 26, 32, 512, 1, 1, False),
    ("ovh 1x1 32->128 @26", 26, 32, 128, 1, 1, False),
    ("ovh 3x3 32->512 @26 res", 26, 32, 512, 3, 1, True),
    # non-power-of-two channel counts: does the pixel pitch matter?
    ("pitch 1x1 1088->512 @13", 13, 1088, 512, 1, 1, False),
    ("pitch 1x1 960->512 @13", 13, 960, 512, 1, 1, False),
    ("pitch 3x3 320->512 @26", 26, 320, 512, 3, 1, True),
    ("pitch 3x3 288->512 @26", 26, 288, 512, 3, 1, True),
    ("pitch 3x3 576->1024 @13", 13, 576, 1024, 3, 1, True),
    ("pitch 1x1 576->256 @26", 26, 576, 256, 1, 1, False),
]


def make_desc(B, H, cin, cout, k, stride, residual, dev):
    pad = k // 2
    Ho = (H + 2 * pad - k) // stride + 1
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn((B, H, H, cin), generator=g).to(dev).to(torch.bfloat16)
    cout_pad = lib.yolo_amd_conv_cout_pad(cout)
    K = k * k * cin
    w = pack_conv_weights((torch.randn((cout_pad, K), generator=g) * (2.0 / K) ** 0.5).to(dev), cout_pad)
    bias = torch.randn(cout_pad, generator=g).to(dev)
    fp32 = cout == 255
    out = torch.empty((B, Ho, Ho, cout), dtype=torch.float32 if fp32 else torch.bfloat16, device=dev)
    res = torch.randn((B, Ho, Ho, cout), generator=g).to(dev).to(torch.bfloat16) if residual else None
    d = (ConvDesc * 1)()
    d[0].src0, d[0].src1 = x.data_ptr(), None
    d[0].weight, d[0].bias = w.data_ptr(), bias.data_ptr()
    d[0].residual = res.data_ptr() if residual else None
    d[0].dst = out.data_ptr()
    d[0].c0, d[0].c1, d[0].up0 = cin, 0, 0
    d[0].B, d[0].H, d[0].W = B, H, H
    d[0].cout, d[0].cout_pad, d[0].ksize, d[0].stride = cout, cout_pad, k, stride
    d[0].k_pad, d[0].leaky, d[0].dst_fp32, d[0].first_layer = K, 1, int(fp32), 0
    flops = 2.0 * B * Ho * Ho * cout * K
    byts = x.numel() * 2 + out.numel() * out.element_size() + (res.numel() * 2 if residual else 0) + cout * K * 2
    return d, (x, w, bias, out, res), flops, byts


def set_variant(v):
    for key in ("YOLO_AMD_CONV_NSTAGE", "YOLO_AMD_CONV_BK", "YOLO_AMD_CONV_TILE", "YOLO_AMD_CONV3",
                "YOLO_AMD_CONV_AMUL", "YOLO_AMD_CONV_BN", "YOLO_AMD_CONV3_WAVES", "YOLO_AMD_CONV3_PRIO", "YOLO_AMD_CONV3_STG",
                "YOLO_AMD_CONV_FP32EPI", "YOLO_AMD_CONV_BN128_MIN", "YOLO_AMD_CONV3_TC", "YOLO_AMD_CONV3_RING", "YOLO_AMD_CONV3_VAR", "YOLO_AMD_CONV3_FLAT", "YOLO_AMD_CONV3_SPLIT"):
        os.environ.pop(key, None)
    for item in filter(None, v.split(",")):
        k, val = item.split("=")
        os.environ[{"ns": "YOLO_AMD_CONV_NSTAGE", "bk": "YOLO_AMD_CONV_BK",
                    "tile": "YOLO_AMD_CONV_TILE", "p3": "YOLO_AMD_CONV3",
                    "am": "YOLO_AMD_CONV_AMUL", "bn": "YOLO_AMD_CONV_BN", "p3w": "YOLO_AMD_CONV3_WAVES", "prio": "YOLO_AMD_CONV3_PRIO", "stg": "YOLO_AMD_CONV3_STG", "f32": "YOLO_AMD_CONV_FP32EPI", "bn128min": "YOLO_AMD_CONV_BN128_MIN", "tc": "YOLO_AMD_CONV3_TC", "ring": "YOLO_AMD_CONV3_RING", "var": "YOLO_AMD_CONV3_VAR", "flat": "YOLO_AMD_CONV3_FLAT", "split": "YOLO_AMD_CONV3_SPLIT"}[k]] = val


def apply_variant(v):
    set_variant(v)
    lib.yolo_amd_reload_knobs()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--variants", default="ns=2;ns=3;ns=4;ns=2,bk=32;ns=4,bk=32;ns=5,bk=32")
    ap.add_argument("--layers", default="")
    ap.add_argument("--zeros", action="store_true", help="zero-filled operands (a clock probe: no data toggling)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    zero = torch.zeros(1024, dtype=torch.uint8, device=dev)
    stream = _lib.current_stream_ptr(dev)
    variants = args.variants.split(";")
    print(f"{'layer':24s} " + " ".join(f"{v:>16s}" for v in variants) + "   (us median | TFLOP/s | GB/s min-traffic)")
    for spec in LAYERS:
        if args.layers and args.layers not in spec[0]:
            continue
        d, keep, flops, byts = make_desc(args.batch, *spec[1:], dev)
        if args.zeros:
            keep[0].zero_(); keep[1].zero_()
            if keep[4] is not None:
                keep[4].zero_()
        times = {v: [] for v in variants}
        for rep in range(args.reps + 3):
            for v in variants:
                apply_variant(v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                check(lib.yolo_amd_conv_forward(d, 1, zero.data_ptr(), stream), "conv")
                e1.record()
                torch.cuda.synchronize()
                if rep >= 3:
                    times[v].append(e0.elapsed_time(e1) * 1e3)
        cells = []
        for v in variants:
            t = float(np.median(times[v]))
            cells.append(f"{t:6.1f}|{flops / t / 1e6:5.0f}|{byts / t / 1e3:4.0f}")
        print(f"{spec[0]:24s} " + " ".join(f"{c:>16s}" for c in cells), flush=True)


if __name__ == "__main__":
    main()
