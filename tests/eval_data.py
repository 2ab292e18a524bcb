"""Shared pieces of the scoring tests: the data-set batches the golden file was made
from and a model stand-in that replays their predictions."""
import json
import os

import torch

import gen_inputs as gi

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "golden_eval.json")) as _fh:
    GOLDEN = json.load(_fh)

# name: (seed, images, batch, N, C, H, confidences) -- as in golden/make_golden_eval.py
CASES = {
    "coco_sweep": (61, 8, 4, 300, 80, 416, [0.2, 0.35, 0.5, 0.65, 0.8]),
    "four_class": (62, 6, 3, 120, 4, 320, [0.25, 0.5]),
    "single_conf": (63, 4, 2, 200, 20, 416, None),
}


def batches(name):
    seed, images, batch, N, C, H, _ = CASES[name]
    out = []
    for k in range(images // batch):
        pred, labels = gi.eval_case(seed + k, batch, N, C, H=H)
        out.append({"image": torch.zeros(batch, 3, H, H), "label": labels, "pred": pred})
    return out


class ReplayModel:
    """Returns the k-th stored prediction at the k-th call, on `device`."""
    def __init__(self, params, preds, device="cpu"):
        self.params, self.preds, self.k, self.device = params, preds, 0, device

    def eval(self):
        pass

    def train(self):
        pass

    def __call__(self, image):
        p = torch.from_numpy(self.preds[self.k]).to(self.device)
        self.k += 1
        return p


def check_against_golden(name, res, train):
    want = GOLDEN[name]["train" if train else "eval"]
    assert float(res[0]) == want["best_map"]
    assert [float(a) for a in res[1]] == want["best_ap"]
    assert float(res[2]) == want["best_conf"]
    assert (None if res[3] is None else float(res[3])) == want["specific_conf_map"]
    assert (None if res[4] is None else [float(a) for a in res[4]]) == want["specific_conf_ap"]
    if not train:
        frame = res[5]
        assert [float(c) for c in frame.columns] == want["frame_columns"]
        for c, col in zip(frame.columns, want["frame"]):
            assert [float(v) for v in frame[c].values] == col
