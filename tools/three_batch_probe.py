"""Throughput with 1, 2, 3 and 4 batches in flight (one HIP stream each).  python tools/three_batch_probe.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

sys.path[:0] = [os.path.join(ROOT, "tests", "golden")]
import gen_inputs as gi  # noqa: E402
from yolo_amd.utilis import filter_results_async  # noqa: E402

H, B, C, A = 416, 32, 80, 3
model, params = bench.build_model(H, C, A, torch.device("cuda"), 0.5, 0.4)[:2]
xs = [torch.from_numpy(gi.images(i, B, H)).cuda() for i in range(4)]
pool = [torch.cuda.Stream() for _ in range(10)]


def run(streams, steps):
    n = len(streams)
    pending = [None] * n
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        s = i % n
        if pending[s] is not None:
            pending[s].result()
        with torch.cuda.stream(streams[s]):
            pending[s] = filter_results_async(model(xs[s]), params)
    for p in pending:
        if p is not None:
            p.result()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for ids in [(0,), (0, 1), (0, 4), (0, 1, 4), (0, 1, 5), (0, 4, 8), (0, 1, 4, 5), (0, 1, 2, 3)]:
    st = [pool[i] for i in ids]
    run(st, 8)
    ms = run(st, 48)
    print(f"streams {ids}: {ms:.3f} ms per batch = {B / ms * 1e3:.0f} images/s", flush=True)
