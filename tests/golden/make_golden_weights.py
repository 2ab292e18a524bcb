"""Pin the Darknet .weights reader against the REFERENCE's own load_weights
(/root/reference/yolo_v3.py:498-617).

Run in the build container only (the reference never travels to the GPU box):

    python tests/golden/make_golden_weights.py

A seeded synthetic .weights file (5 x int32 header, then per convolution
[bn.bias, bn.weight, running_mean, running_var | conv.bias], conv.weight, fp32)
is written with weights_file_bytes() below, the reference model reads it with
and without cust_train_zero, and a sha256 of every state_dict tensor it ends up
with is stored in golden_weights.json.  The test regenerates the same file from
the same seed and requires yolo_amd.yolo_v3.load_weights to reproduce every
digest -- including the cust_train_zero pointer quirk (the reference skips a
constant 255 values for the zero-filled bias AND for the zero-filled weight
tensor of the three detection convolutions, so every later layer reads shifted
data).
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import gen_inputs as gi  # noqa: E402

CASES = [
    # name, C, A, seed
    ("coco_c80_a3", 80, 3, 7),
    ("one_anchor_c4_a1", 4, 1, 8),
]


def weights_file_bytes(table, seed, seen=123):
    """The synthetic file: gen_inputs.synthetic_state_dict in Darknet order."""
    sd = gi.synthetic_state_dict(table, seed)
    chunks = [np.array([0, 2, 0, seen, 0], dtype=np.int32).tobytes()]
    for i, ent in enumerate(table):
        if ent["type"] != "conv":
            continue
        if ent["bn"]:
            p = f"module_list.{i}.bach_norm_{i}."
            for k in ("bias", "weight", "running_mean", "running_var"):
                chunks.append(sd[p + k].tobytes())
        else:
            chunks.append(sd[f"module_list.{i}.conv_{i}.bias"].tobytes())
        chunks.append(sd[f"module_list.{i}.conv_{i}.weight"].tobytes())
    return b"".join(chunks)


def digest_state_dict(sd):
    """{key: first 16 hex digits of the sha256 of the tensor's bytes} (the counters
    num_batches_tracked are left out: loading a file does not touch them)."""
    out = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            continue
        out[k] = hashlib.sha256(np.ascontiguousarray(v.detach().cpu().numpy()).tobytes()).hexdigest()[:16]
    return out


def main():
    from make_golden import import_reference
    from yolo_amd.cfg import default_params, write_yolov3_cfg
    import yolo_oracle as orc
    ref_utilis, ref_yolo = import_reference()
    golden = {"torch": torch.__version__, "cases": {}}
    for name, C, A, seed in CASES:
        cfg = write_yolov3_cfg(os.path.join(HERE, "_tmp_w.cfg"), C, A)
        table = orc.layer_table(orc.parse_cfg(cfg), 3, C, A)
        path = os.path.join(HERE, "_tmp_w.weights")
        with open(path, "wb") as fh:
            fh.write(weights_file_bytes(table, seed))
        entry = {"C": C, "A": A, "seed": seed, "file_bytes": os.path.getsize(path)}
        for mode in (False, True):
            params = default_params(416, C, A)
            params["cuda"] = False
            blocks = ref_utilis.parse_cfg(cfg)
            orig = ref_yolo.yolo_v3.load_weights
            ref_yolo.yolo_v3.load_weights = lambda self, *a, **k: None
            try:
                torch.manual_seed(0)
                model = ref_yolo.yolo_v3(params, blocks)
            finally:
                ref_yolo.yolo_v3.load_weights = orig
            model.load_weights(path, cust_train_zero=mode)
            entry["zero" if mode else "plain"] = {
                "seen": int(model.seen), "header": model.header.tolist(),
                "digests": digest_state_dict(model.state_dict())}
            print(name, "cust_train_zero =", mode, len(entry["zero" if mode else "plain"]["digests"]), "tensors",
                  flush=True)
        golden["cases"][name] = entry
        os.remove(path)
        os.remove(cfg)
    with open(os.path.join(HERE, "golden_weights.json"), "w") as fh:
        json.dump(golden, fh, indent=0)
    print("wrote golden_weights.json")


if __name__ == "__main__":
    main()
