"""Fifty batch-1 forward passes plus detection filtering, for a kernel trace of the video
configuration.  rocprofv3 --kernel-trace --stats -d OUT -- python3 tools/latency_run.py [H]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

sys.path[:0] = [os.path.join(ROOT, "tests", "golden")]
import gen_inputs as gi  # noqa: E402
from yolo_amd import utilis  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 320
model, params = bench.build_model(H, 80, 3, torch.device("cuda"), 0.5, 0.4)[:2]
x = torch.from_numpy(gi.images(1, 1, H)).cuda()
for _ in range(50):
    utilis.filter_results(model(x), params)
torch.cuda.synchronize()
