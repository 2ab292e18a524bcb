"""Latency of the batch-1 forward pass launched kernel by kernel against the same launches replayed
from a captured HIP graph.  python tools/graph_probe.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

sys.path[:0] = [os.path.join(ROOT, "tests", "golden")]
import gen_inputs as gi  # noqa: E402

for H in (320, 416):
    model, params = bench.build_model(H, 80, 3, torch.device("cuda"), 0.5, 0.4)[:2]
    x = torch.from_numpy(gi.images(1, 1, H)).cuda()

    def timed(fn, reps=200):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    eager = timed(lambda: model(x))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            ref = model(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            y = model(x)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    same = bool(torch.equal(y, ref))
    graph = timed(g.replay)
    print(f"{H} / batch 1 forward: eager {eager:.3f} ms, graph replay {graph:.3f} ms, same output: {same}", flush=True)
