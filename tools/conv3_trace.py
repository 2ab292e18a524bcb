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
    trace = torch.zeros(1 << 18, dtype=torch.int64, device=dev)
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
    raw = trace.cpu().numpy().reshape(-1, 8)
    raw = raw[raw[:, 0] > 0]
    t = raw[:, :4].astype(np.float64) / 100.0               # microseconds
    t0 = t[:, 0].min()
    t -= t0
    n = t.shape[0]
    print(f"{a.layer}, batch {a.batch}: {n} workgroups, kernel {e0.elapsed_time(e1) * 1e3:.1f} us by events, "
          f"{t[:, 3].max():.1f} us first start -> last stores issued")
    full = trace.cpu().numpy().reshape(-1, 8)
    idx = np.nonzero(full[:, 0] > 0)[0]
    kloop = t[:, 2] - t[:, 1]
    print("  K loop by blockIdx % 8 (workgroups that share an XCD):",
          " ".join(f"{kloop[idx % 8 == x].mean():.1f}" for x in range(8)))
    print("  K loop by launch order, eighths of the grid:",
          " ".join(f"{kloop[(idx * 8) // (idx.max() + 1) == e].mean():.1f}" for e in range(8)))
    print("  K loop by start time, quartiles of the first 512:",
          " ".join(f"{q:.1f}" for q in np.quantile(kloop[np.argsort(t[:, 0])[:512]], [0, 0.25, 0.5, 0.75, 1])))
    # which workgroups share a CU: XCC_ID (high word) and, in HW_ID, CU_ID bits 11:8, SH bit 12, SE bits 15:13
    hw = raw[:, 4]
    cu_key = ((hw >> 32) << 16) | (hw & 0xFF00)
    keys, inv, counts = np.unique(cu_key, return_inverse=True, return_counts=True)
    print(f"  {len(keys)} distinct (XCC, SE, SH, CU) places; workgroups per place: min {counts.min()} max {counts.max()}")
    first_on_cu = np.zeros(len(t), dtype=bool)
    for k in range(len(keys)):
        members = np.nonzero(inv == k)[0]
        first_on_cu[members[np.argmin(t[members, 0])]] = True
    print(f"  K loop of the workgroup that started first on its CU: {kloop[first_on_cu].mean():.1f}, of the others: "
          f"{kloop[~first_on_cu].mean():.1f}")
    # turnover on a CU: what lies between one workgroup's stores and its successor's K loop
    life = t[:, 3] - t[:, 0]
    print(f"  all workgroups: set-up {np.mean(t[:, 1] - t[:, 0]):.2f}  K loop {kloop.mean():.2f}  "
          f"epilogue {np.mean(t[:, 3] - t[:, 2]):.2f}  lifetime {life.mean():.2f} us")
    if raw[:, 5].max() > 0:
        t5 = raw[:, 5].astype(np.float64) / 100.0 - t0
        t6 = raw[:, 6].astype(np.float64) / 100.0 - t0
        print(f"  epilogue: K loop done -> residual tile in LDS {np.mean(t5 - t[:, 2]):.2f}, -> output tile in LDS "
              f"{np.mean(t6 - t5):.2f}, -> stores issued {np.mean(t[:, 3] - t6):.2f} us")
    gaps, busy2 = [], []
    for k in range(len(keys)):
        members = np.nonzero(inv == k)[0]
        members = members[np.argsort(t[members, 0])]
        ends = np.sort(t[members, 3])
        # the i-th start on this CU (i >= 2) follows the (i-2)-th end: two slots per CU
        for i in range(2, len(members)):
            gaps.append(t[members[i], 0] - ends[i - 2])
        span = t[members, 3].max() - t[members, 0].min()
        busy2.append(np.sum(t[members, 2] - t[members, 1]) / (2 * span) if span > 0 else 0)
    if gaps:
        print(f"  successor starts {np.mean(gaps):.2f} us after a slot's stores were issued (median {np.median(gaps):.2f}); "
              f"share of the two slots' time spent inside K loops: {np.mean(busy2):.2f}")
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
