"""Shared test plumbing: golden fixtures, the C oracle (ctypes) and small
comparison helpers.  Test infrastructure only."""
from __future__ import annotations

import ctypes
import hashlib
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
ORACLE_DIR = os.path.join(ROOT, "oracle")


def load_golden():
    with open(os.path.join(GOLDEN_DIR, "golden_meta.json")) as fh:
        meta = json.load(fh)
    arrays = dict(np.load(os.path.join(GOLDEN_DIR, "golden_arrays.npz")))
    return meta, arrays


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


_ORACLE_LIB = None


def oracle_lib():
    """ctypes handle on oracle/liboracle.so, building it with gcc if needed."""
    global _ORACLE_LIB
    if _ORACLE_LIB is not None:
        return _ORACLE_LIB
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    src = os.path.join(ORACLE_DIR, "nms_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"])
    lib = ctypes.CDLL(so)
    fp = ctypes.POINTER(ctypes.c_float)
    lib.oracle_filter_results.restype = ctypes.c_long
    lib.oracle_filter_results.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_float, ctypes.c_float, fp]
    lib.oracle_box_iou.restype = None
    lib.oracle_box_iou.argtypes = [fp, ctypes.c_int, fp, ctypes.c_int, fp]
    lib.oracle_decode_head.restype = None
    lib.oracle_decode_head.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, fp, ctypes.c_float, fp]
    _ORACLE_LIB = lib
    return lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def c_filter_results(pred, conf, nms, C):
    """C oracle; returns (K,9) float32 array (K may be 0)."""
    pred = np.ascontiguousarray(pred, dtype=np.float32)
    B, N, D = pred.shape
    out = np.empty((max(B * N, 1), 9), dtype=np.float32)
    K = oracle_lib().oracle_filter_results(_fp(pred), B, N, D, C, conf, nms, _fp(out))
    return out[:K].copy()


def c_box_iou(b1, b2):
    b1 = np.ascontiguousarray(b1, dtype=np.float32)
    b2 = np.ascontiguousarray(b2, dtype=np.float32)
    out = np.empty(b2.shape[0], dtype=np.float32)
    oracle_lib().oracle_box_iou(_fp(b1), b1.shape[0], _fp(b2), b2.shape[0], _fp(out))
    return out


def c_decode_head(raw, head_anchors, H, C):
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    B, ch, G, _ = raw.shape
    D = 5 + C
    A = ch // D
    stride = H // G
    anc = np.array([[np.float32(a[0] / stride), np.float32(a[1] / stride)]
                    for a in head_anchors], dtype=np.float32)
    out = np.empty((B, G * G * A, D), dtype=np.float32)
    oracle_lib().oracle_decode_head(_fp(raw), B, A, D, G, _fp(anc), float(stride), _fp(out))
    return out


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    same = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))
    if not same.all():
        bad = np.argwhere(~same)
        raise AssertionError(f"{what}: {len(bad)} of {a.size} values differ; first at "
                             f"{bad[0].tolist()}: {a[tuple(bad[0])]!r} vs {b[tuple(bad[0])]!r}")


def nms_case_input(case):
    import gen_inputs as gi
    pred = gi.nms_pred(**case["gen"])
    if case.get("edit"):
        pred = gi.EDITS[case["edit"]](pred)
    return pred
