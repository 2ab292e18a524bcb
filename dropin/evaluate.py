"""``from evaluate import get_map`` (train.py, yolo_v3.fit callers) resolves to the
MI355X implementation when this directory is ahead of the reference checkout on
sys.path."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))

from yolo_amd.evaluate import bbox_iou_numpy, compute_map, get_map  # noqa: F401,E402
