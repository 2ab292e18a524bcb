"""``from generate_anchors import set_anchors_to_model`` resolves to the MI355X
package's implementation when this directory is ahead of the reference checkout on
sys.path."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))

from yolo_amd.generate_anchors import (IOU, avg_IOU, generate_anchor, get_anchors,  # noqa: F401,E402
                                       kmeans, set_anchors_to_model, write_anchors_to_file)
