"""Put this directory on sys.path ahead of the reference checkout and
``from utilis import ...`` in detection.py / video.py / get_fps.py /
evaluate.py resolves to the MI355X implementation."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))

from yolo_amd.utilis import *  # noqa: F401,F403,E402
from yolo_amd.utilis import _save_checkpoint  # noqa: F401,E402
