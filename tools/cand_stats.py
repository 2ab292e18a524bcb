"""Candidate-list sizes of the benchmark's NMS input: per image and class group (the
sort kernel's workgroups) and the largest class segments.  python tools/cand_stats.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

sys.path[:0] = [os.path.join(ROOT, "tests", "golden")]
import gen_inputs as gi  # noqa: E402

H, B, C, A = 416, 32, 80, 3
model, params = bench.build_model(H, C, A, torch.device("cuda"), 0.5, 0.4)[:2]
x = torch.from_numpy(gi.images(1, B, H)).cuda()
pred = model(x)
obj = pred[..., 4] > 0.5
cls = pred[..., 5:].argmax(-1)
grp = (cls * 8) // C
per_list = torch.stack([((grp == g) & obj).sum(1) for g in range(8)], 1).cpu().numpy()
print("candidates per image: min %d mean %.0f max %d" % (obj.sum(1).min(), obj.sum(1).float().mean(), obj.sum(1).max()))
print("per (image, class group) list: mean %.0f max %d; lists > 1024: %d, > 2048: %d of %d"
      % (per_list.mean(), per_list.max(), (per_list > 1024).sum(), (per_list > 2048).sum(), per_list.size))
seg = torch.stack([((cls == c) & obj).sum(1) for c in range(C)], 1).cpu().numpy()
print("largest class segments:", np.sort(seg.reshape(-1))[-5:])
