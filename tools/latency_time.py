"""Batch-1 latency (forward pass + detection filtering, one frame after the other) at the video
sizes.  python tools/latency_time.py [sizes...]; knobs come from the environment."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

sys.path[:0] = [os.path.join(ROOT, "tests", "golden")]
import gen_inputs as gi  # noqa: E402
from yolo_amd import utilis  # noqa: E402

sizes = [tuple(int(v) for v in (s + "x1").split("x")[:2]) for s in sys.argv[1:]] or [(320, 1), (416, 1)]
out = []
for H, B in sizes:
    model, params = bench.build_model(H, 80, 3, torch.device("cuda"), 0.5, 0.4)[:2]
    x = torch.from_numpy(gi.images(1, B, H)).cuda()

    def frame():
        return utilis.filter_results(model(x), params)

    def fwd():
        return model(x)

    res = []
    for fn in (fwd, frame):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(100):
                fn()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 100 * 1e3)
        res.append(best)
    out.append(f"{H}x{B}: forward {res[0]:.3f} ms, frame {res[1]:.3f} ms")
print(" | ".join(out), flush=True)
