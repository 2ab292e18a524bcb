"""Phase timeline of the 3x3 patch kernel: every workgroup stamps four 100 MHz
timestamps (start, K loop entered, K loop done, stores issued) into a buffer whose
address is handed over in YOLO_AMD_CONV3_TRACE (development aid of the library).
    python tools/conv3_trace.py [--layer "3x3 256->512 @26"] [--batch 32]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import conv_bench as cb  # noqa: E402
from yolo_amd import _lib  # noqa: E402
from yolo_amd._lib import check, lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", default="3x3 256->512 @26")
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    spec = [s for s in cb.LAYERS if s[0] == a.layer][0]
    d, keep, flops, byts = cb.make_desc(a.batch, *spec[1:], dev)
    zero = torch.zeros(1024, dtype=torch.uint8, device=dev)
    stream = _lib.current_stream_ptr(dev)
    trace = torch.zeros(1 << 16, dtype=torch.int64, device=dev)
    for _ in range(3):
        check(lib.yolo_amd_conv_forward(d, 1, zero.data_ptr(), stream), "conv")
    os.environ["YOLO_AMD_CONV3_TRACE"] = hex(trace.data_ptr())
    lib.yolo_amd_reload_knobs()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.yolo_amd_conv_forward(d, 1, zero.data_ptr(), stream), "conv")
    e1.record()
    torch.cuda.synchronize()
    del os.environ["YOLO_AMD_CONV3_TRACE"]
    lib.yolo_amd_reload_knobs()
    t = trace.cpu().numpy().reshape(-1, 4)
    t = t[t[:, 0] > 0].astype(np.float64) / 100.0          # microseconds
    t0 = t[:, 0].min()
    t -= t0
    n = t.shape[0]
    print(f"{a.layer}, batch {a.batch}: {n} workgroups, kernel {e0.elapsed_time(e1) * 1e3:.1f} us by events, "
          f"{t[:, 3].max():.1f} us first start -> last stores issued")
    order = np.argsort(t[:, 0])
    first_round = t[order[: min(n, 512)]]
    later = t[order[512:]] if n > 512 else None
    for name, grp in (("first 512 workgroups", first_round), ("the rest", later)):
        if grp is None or not len(grp):
            continue
        print(f"  {name}: start {grp[:, 0].mean():6.2f} (max {grp[:, 0].max():6.2f})  "
              f"set-up {np.mean(grp[:, 1] - grp[:, 0]):5.2f}  K loop {np.mean(grp[:, 2] - grp[:, 1]):6.2f} "
              f"(min {np.min(grp[:, 2] - grp[:, 1]):6.2f} max {np.max(grp[:, 2] - grp[:, 1]):6.2f})  "
              f"epilogue {np.mean(grp[:, 3] - grp[:, 2]):5.2f}  end {grp[:, 3].mean():6.2f} (max {grp[:, 3].max():6.2f})")


if __name__ == "__main__":
    main()
