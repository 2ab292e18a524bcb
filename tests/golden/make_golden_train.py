"""Golden vectors for the model in train() mode -- the mode the callers that never call
eval() run it in (/root/reference/video.py:21-25, get_fps.py:25-29): BatchNorm2d on batch
statistics, running statistics moved by every forward call.

Run in the build container only (imports the REFERENCE, /root/reference/yolo_v3.py):

    python tests/golden/make_golden_train.py

Writes golden_train.npz (outputs) and golden_train.json (digests of the running statistics
after two calls).  Inputs and weights are the seeded ones of gen_inputs.py.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_inputs as gi  # noqa: E402
from make_golden import build_ref_model, import_reference  # noqa: E402
from yolo_amd.cfg import default_params, write_yolov3_cfg  # noqa: E402

CASES = [("train_h64_b2", 64, 2, 80, 3), ("train_h96_b1", 96, 1, 80, 3), ("train_h64_a1c4", 64, 3, 4, 1)]
IMAGE_SEEDS = (5, 6)


def stat_digest(t):
    a = t.detach().double().numpy().reshape(-1)
    idx = np.linspace(0, a.size - 1, min(4, a.size)).astype(np.int64)
    return dict(sum=float(a.sum()), abssum=float(np.abs(a).sum()), samples=a[idx].tolist())


def main():
    ref_utilis, ref_yolo = import_reference()
    arrays, meta = {}, {"torch": torch.__version__, "image_seeds": list(IMAGE_SEEDS), "weight_seed": 3, "cases": []}
    for name, H, B, C, A in CASES:
        p = default_params(H, C, A)
        p["cuda"] = False
        cfgp = write_yolov3_cfg(os.path.join(HERE, "_tmp_yolov3_train.cfg"), C, A)
        model, _ = build_ref_model(ref_utilis, ref_yolo, p, cfgp, seed=3)
        os.remove(cfgp)
        model.train()                      # what a freshly constructed model is
        outs = []
        for s in IMAGE_SEEDS:
            with torch.no_grad():
                outs.append(model(torch.from_numpy(gi.images(s, B, H))).numpy().astype(np.float32))
        case = dict(name=name, H=H, B=B, C=C, A=A, out_shape=list(outs[0].shape), bn={})
        for k, y in enumerate(outs):
            arrays[f"{name}_call{k}"] = y
        sd = model.state_dict()
        for key, t in sd.items():
            if "bach_norm" in key and key.endswith(("running_mean", "running_var")):
                case["bn"][key] = stat_digest(t)
            elif key.endswith("num_batches_tracked"):
                assert int(t) == len(IMAGE_SEEDS)
        case["num_batches_tracked"] = len(IMAGE_SEEDS)
        # ... and the moved statistics as the eval() model then uses them
        model.eval()
        with torch.no_grad():
            arrays[f"{name}_eval_after"] = model(torch.from_numpy(gi.images(IMAGE_SEEDS[0], B, H))).numpy().astype(np.float32)
        meta["cases"].append(case)
        print(name, [o.shape for o in outs], "obj range", float(outs[0][..., 4].min()), float(outs[0][..., 4].max()))
    np.savez_compressed(os.path.join(HERE, "golden_train.npz"), **arrays)
    with open(os.path.join(HERE, "golden_train.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    print("wrote", sum(a.nbytes for a in arrays.values()) / 1e6, "MB raw")


if __name__ == "__main__":
    main()
