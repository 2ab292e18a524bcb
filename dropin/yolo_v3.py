"""See utilis.py in this directory: ``from yolo_v3 import yolo_v3`` resolves to
the MI355X implementation."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))

from yolo_amd.yolo_v3 import yolo_v3, Darknet, decode_heads  # noqa: F401,E402
