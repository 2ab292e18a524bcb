"""Golden vectors for the anchor k-means (/root/reference/generate_anchors.py),
produced by the REFERENCE's own functions on seeded label sizes.  Build container
only:   python tests/golden/make_golden_anchors.py

The reference spells float64 as `np.float`, an alias numpy 2 no longer has; the
alias is restored on the numpy module for the duration of this script so that the
reference's kmeans runs unmodified.
"""
import contextlib
import io
import json
import os
import random
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_inputs as gi  # noqa: E402

CASES = {"nine": (71, 900, 9, 416.0, 416.0), "three": (72, 300, 3, 512.0, 320.0),
         "six": (73, 500, 6, 608.0, 608.0)}


def main():
    np.float = float
    sys.path.insert(0, "/root/reference")
    import generate_anchors as ref
    out = {}
    for name, (seed, n, k, w, h) in CASES.items():
        dims = gi.label_dims(seed, n)
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "label.csv")
            gi.write_label_csv(path, dims, seed)
            random.seed(seed)
            with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
                anchors = ref.generate_anchor(path, w, h, num_clusters=k)
        cent = anchors / np.array([w, h])
        out[name] = {"anchors": anchors.tolist(),
                     "avg_iou": float(ref.avg_IOU(dims, cent)),
                     "iou_row0": ref.IOU(dims[0], cent).tolist(),
                     "iou_row3": ref.IOU(dims[3], dims[:8]).tolist()}
        print(name, np.round(anchors, 1).tolist())
    with open(os.path.join(HERE, "golden_anchors.json"), "w") as fh:
        json.dump(out, fh)


if __name__ == "__main__":
    main()
