"""Where a wave of the flat-patch 3x3 kernel spends a tap: the trace build of the library
(make -C yolo_amd/csrc trace) adds up, per wave, the shader cycles between five points of the
tap loop; this prints their averages per tap.
    YOLO_AMD_LIB=yolo_amd/libyolo_amd_trace.so YOLO_AMD_CONV3_FLAT=1 python tools/tap_trace.py [--layer "3x3 256->512 @26"]"""
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

NAMES = ["burst end -> next tap", "wait for weights", "issue LDS-DMA", "fragment reads, issued -> back",
         "MFMA burst", "barrier (once per nine taps)"]


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
    full = trace.cpu().numpy()
    wg = full[:65536].reshape(-1, 8)
    nwg = int((wg[:, 0] > 0).sum())
    kloop_us = (wg[:nwg, 2] - wg[:nwg, 1]) / 100.0
    seg = full[65536:65536 + nwg * 64].reshape(nwg, 8, 8)[:, :, :6].astype(np.float64)
    taps = spec[2] // 32 * 9
    per_tap = seg / taps
    tot = per_tap.sum(axis=2)
    print(f"{a.layer}, batch {a.batch}: {nwg} workgroups, {e0.elapsed_time(e1) * 1e3:.1f} us with the stamps in; "
          f"K loop {kloop_us.mean():.1f} us = {taps} taps of {kloop_us.mean() / taps * 1e3:.0f} ns; "
          f"{tot.mean():.0f} s_memtime ticks per tap and wave ({tot.mean() / (kloop_us.mean() / taps * 1e3):.2f} per ns)")
    # workgroup phases (wave 0's stamps, 10 ns ticks): start, first loads issued, K loop done, pieces staged, end
    ph = wg[:nwg].astype(np.float64)
    t_first, t_last = ph[:, 0].min(), ph[:, 3].max()
    print(f"  workgroup phases, us (mean): set-up {(ph[:, 1] - ph[:, 0]).mean() / 100:.2f}; K loop {(ph[:, 2] - ph[:, 1]).mean() / 100:.2f}; "
          f"epilogue to pieces staged {(ph[:, 6] - ph[:, 2]).mean() / 100:.2f}; stores issued {(ph[:, 3] - ph[:, 6]).mean() / 100:.2f}; "
          f"alive {(ph[:, 3] - ph[:, 0]).mean() / 100:.2f}; first start to last end {(t_last - t_first) / 100:.1f}; "
          f"starts within the first us {int(((ph[:, 0] - t_first) < 100).sum())}")
    order = np.sort(ph[:, 0] - t_first) / 100
    print("  start times, us, deciles: " + " ".join(f"{order[int(k * (nwg - 1) / 10)]:.1f}" for k in range(11)))
    for half, nm in ((slice(0, 4), "waves 0-3 (the larger share of the pixel groups)"), (slice(4, 8), "waves 4-7 (the smaller share)")):
        m = per_tap[:, half, :].mean(axis=(0, 1))
        print(f"  {nm}: " + "; ".join(f"{NAMES[i]} {m[i]:.0f}" for i in range(6)) + f"; sum {m.sum():.0f}")
    q = np.quantile(per_tap.reshape(-1, 6), [0.1, 0.5, 0.9], axis=0)
    for i in range(6):
        print(f"  {NAMES[i]:32s} p10 {q[0, i]:6.0f}  median {q[1, i]:6.0f}  p90 {q[2, i]:6.0f} ticks per tap")


if __name__ == "__main__":
    main()
