"""Throughput with two batches in flight: batch i on one HIP stream, batch i+1 on another,
each running the serial chain model(x) -> filter_results.  python tools/two_batch_probe.py"""
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
xs = [torch.from_numpy(gi.images(i, B, H)).cuda() for i in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(n_streams, steps):
    pending = [None] * n_streams
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        s = i % n_streams
        if pending[s] is not None:
            pending[s].result()
        with torch.cuda.stream(streams[s]):
            pending[s] = filter_results_async(model(xs[s]), params)
    for p in pending:
        if p is not None:
            p.result()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for n in (1, 2):
    run(n, 6)
    ms = run(n, 40)
    print(f"{n} stream(s): {ms:.3f} ms per batch of {B} = {B / ms * 1e3:.0f} images/s", flush=True)


def run_variant(name, steps, wait_main=False, record=False):
    pending = [None, None]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        s = i % 2
        if pending[s] is not None:
            pending[s].result()
        if wait_main:
            streams[s].wait_stream(torch.cuda.current_stream())
        if record:
            xs[s].record_stream(streams[s])
        with torch.cuda.stream(streams[s]):
            pending[s] = filter_results_async(model(xs[s]), params)
    for p in pending:
        if p is not None:
            p.result()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for name, kw in [("plain", {}), ("wait_stream(main)", dict(wait_main=True)), ("record_stream", dict(record=True))]:
    run_variant(name, 6, **kw)
    ms = run_variant(name, 40, **kw)
    print(f"two streams, {name}: {ms:.3f} ms per batch = {B / ms * 1e3:.0f} images/s", flush=True)

from yolo_amd.harness import DetectionPipeline  # noqa: E402
pipe = DetectionPipeline(model, params)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(40):
        pipe.submit(xs[i % 2])
    pipe.flush()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 40 * 1e3
print(f"DetectionPipeline: {ms:.3f} ms per batch = {B / ms * 1e3:.0f} images/s", flush=True)

more = [torch.cuda.Stream() for _ in range(6)]
allst = streams + list(pipe.lanes) + more
print("stream handles:", [hex(s.cuda_stream) for s in allst])
for a, b in [(2, 3), (4, 5), (0, 2), (0, 3), (1, 2), (4, 6), (0, 4)]:
    streams[0], streams[1] = allst[a], allst[b]
    run_variant("pair", 6)
    ms = run_variant("pair", 30)
    print(f"streams #{a} + #{b}: {ms:.3f} ms per batch = {B / ms * 1e3:.0f} images/s", flush=True)
