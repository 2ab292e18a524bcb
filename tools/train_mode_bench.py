"""Latency of model(x) in eval() mode, in train() mode on the HIP path (batch-statistics
BatchNorm) and through the torch modules, for the video callers' shape (320, batch 1) and
a batch of 416 images.  Run on a GPU box: python tools/train_mode_bench.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "oracle")]
import gen_inputs as gi  # noqa: E402
from gpu_helpers import build_model  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for H, B, reps in [(320, 1, 50), (416, 1, 50), (416, 16, 20)]:
    model, params, _ = build_model(H)
    x = torch.from_numpy(gi.images(3, B, H)).cuda()
    with torch.no_grad():
        ev = timed(lambda: model(x), reps)
        model.train()
        tr = timed(lambda: model(x), reps)
        mod = timed(lambda: model._forward_modules(x), max(reps // 5, 3))
    print(f"{H} / batch {B}: eval {ev:.3f} ms, train (HIP) {tr:.3f} ms, train (torch modules) {mod:.3f} ms", flush=True)
