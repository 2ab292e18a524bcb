"""Cases of golden/golden_pre.npz (see golden/make_golden_pre.py) for the CPU and
GPU tests of the image transforms."""
import hashlib
import os

import numpy as np

import gen_inputs as gi

PRE = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_pre.npz"))
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
CASES = {
    "landscape_416": (81, 416, 416, 234, 0, 256),
    "portrait_96": (82, 96, 40, 96, 0, 256),
    "full_64_narrow_range": (83, 64, 64, 64, 30, 200),
    "dark_32": (84, 32, 32, 20, 140, 250),
}
MODES = {"minmax": dict(minmax=True), "detection": dict(minmax=True, mean=MEAN, std=STD),
         "prep": dict(minmax=False, div255=True)}


def resized_image(name):
    seed, side, rw, rh, lo, hi = CASES[name]
    return gi.photo(seed, rw, rh, lo, hi), side


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def check_against_golden(name, key, got):
    side = CASES[name][1]
    assert got.dtype == np.float32 and got.shape == (3, side, side)
    step = max(1, side // 8)
    np.testing.assert_array_equal(got[:, ::step, ::step].view(np.int32),
                                  PRE[f"{name}_{key}_sample"].view(np.int32))
    if f"{name}_{key}" in PRE:
        np.testing.assert_array_equal(got.view(np.int32), PRE[f"{name}_{key}"].view(np.int32))
    assert sha(got) == str(PRE[f"{name}_{key}_sha"])
