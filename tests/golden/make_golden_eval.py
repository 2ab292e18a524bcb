"""Golden vectors for the scoring code (/root/reference/evaluate.py: bbox_iou_numpy,
compute_map, get_map), produced by running the REFERENCE's own functions on the
seeded data sets of gen_inputs.eval_case.  Run in the build container only:

    python tests/golden/make_golden_eval.py

evaluate.py imports torchvision and the data-set module (cv2, skimage) at module
level for code that get_map never calls; empty stand-ins are registered for them,
as make_golden.py does.  The model handed to the reference's get_map is a stand-in
that returns the seeded decoded predictions batch by batch, so the reference runs
its own filter_results, matching and AP arithmetic on them.
"""
from __future__ import annotations

import contextlib
import io
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import gen_inputs as gi  # noqa: E402
from yolo_amd.cfg import default_params  # noqa: E402

CASES = {
    # name: (seed, images, batch, N, C, H, conf list or None for the single specific_conf)
    "coco_sweep": (61, 8, 4, 300, 80, 416, [0.2, 0.35, 0.5, 0.65, 0.8]),
    "four_class": (62, 6, 3, 120, 4, 320, [0.25, 0.5]),
    "single_conf": (63, 4, 2, 200, 20, 416, None),
}


def import_reference():
    for name in ["cv2", "tensorboardX", "torchvision", "torchvision.transforms",
                 "skimage", "skimage.color", "data"]:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["tensorboardX"].SummaryWriter = object
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["skimage"].color = sys.modules["skimage.color"]
    sys.modules["data"].CustData = object
    sys.modules["data"].RandomCrop = object
    sys.path.insert(0, "/root/reference")
    import evaluate as ref_eval
    return ref_eval


def batches(seed, images, batch, N, C, H):
    out = []
    for k in range(images // batch):
        pred, labels = gi.eval_case(seed + k, batch, N, C, H=H)
        out.append({"image": torch.zeros(batch, 3, H, H), "label": labels, "pred": pred})
    return out


class ReplayModel:
    """Returns the k-th stored prediction at the k-th call."""
    def __init__(self, params, preds):
        self.params, self.preds, self.k = params, preds, 0

    def eval(self):
        pass

    def train(self):
        pass

    def __call__(self, image):
        p = torch.from_numpy(self.preds[self.k])
        self.k += 1
        return p


def main():
    ref = import_reference()
    out = {}
    for name, (seed, images, batch, N, C, H, confs) in CASES.items():
        data = batches(seed, images, batch, N, C, H)
        entry = {}
        for train in (False, True):
            params = default_params(height=H, num_classes=C)
            params["specific_conf"] = 0.5
            model = ReplayModel(params, [d["pred"] for d in data])
            with contextlib.redirect_stdout(io.StringIO()):
                res = ref.get_map(model, data, train=train,
                                  confidence=confs if confs else False)
            rec = {"best_map": float(res[0]), "best_ap": [float(a) for a in res[1]],
                   "best_conf": float(res[2]),
                   "specific_conf_map": None if res[3] is None else float(res[3]),
                   "specific_conf_ap": None if res[4] is None else [float(a) for a in res[4]]}
            if not train:
                frame = res[5]
                rec["frame_columns"] = [float(c) for c in frame.columns]
                rec["frame"] = [[float(v) for v in frame[c].values] for c in frame.columns]
            entry["train" if train else "eval"] = rec
        out[name] = entry
        print(name, entry["eval"]["best_map"], entry["eval"]["best_conf"])
    a, b = gi.iou_pairs(64, m=64)
    out["bbox_iou_numpy"] = {"sha_note": "iou_pairs(64, m=64), full (64, 64) matrix",
                             "matrix": ref.bbox_iou_numpy(a.astype(np.float64),
                                                          b.astype(np.float64)).tolist()}
    with open(os.path.join(HERE, "golden_eval.json"), "w") as fh:
        json.dump(out, fh)
    print("wrote golden_eval.json")


if __name__ == "__main__":
    main()
