"""Experiment: does running the conv stack as two half batches on two HIP streams
(two model copies, so two plans and arenas) beat one full batch?  Tails and
epilogues of one half could fill with the other half's tiles.
    python tools/two_stream_probe.py [--batch 32] [--height 416]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench  # noqa: E402


def timed(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--height", type=int, default=416)
    ap.add_argument("--parts", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    models = [bench.build_model(a.height, 80, 3, dev, 0.5, 0.4)[0] for _ in range(a.parts)]
    x = torch.randn(a.batch, 3, a.height, a.height, device=dev)
    parts = list(x.chunk(a.parts))
    streams = [torch.cuda.Stream() for _ in range(a.parts)]

    def whole():
        with torch.no_grad():
            models[0](x)

    def split():
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(cur)
        with torch.no_grad():
            for m, p, s in zip(models, parts, streams):
                s.wait_event(ev)
                with torch.cuda.stream(s):
                    m(p)
        for s in streams:
            cur.wait_stream(s)

    def serial_parts():
        with torch.no_grad():
            for m, p in zip(models, parts):
                m(p)

    t0 = timed(whole)
    t1 = timed(split)
    t2 = timed(serial_parts)
    print(f"batch {a.batch} @ {a.height}: one plan {t0:.3f} ms, {a.parts} streams {t1:.3f} ms, "
          f"{a.parts} parts serially {t2:.3f} ms")


if __name__ == "__main__":
    main()
