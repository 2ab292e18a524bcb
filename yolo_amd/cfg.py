"""Generator for the cfg text the reference code expects.

The reference ships no cfg file.  Its code implies the layout: 107 blocks,
NO ``[net]`` block (``create_module`` treats ``blocks[0]`` as layer 0,
/root/reference/utilis.py:262), detection convs at block indices 81/93/105
(/root/reference/yolo_v3.py:20), routes whose first item is relative and
second item absolute (/root/reference/utilis.py:206-226) and shortcuts with a
relative ``from`` (/root/reference/utilis.py:229-236).

``yolov3_cfg_text()`` emits exactly that Darknet-53 + 3-head YOLOv3 graph.
"""
from __future__ import annotations

import os
import tempfile

# Stock COCO anchors, largest first: the reference hands anchors[0:A] to the
# FIRST yolo block in file order, which is the stride-32 head
# (/root/reference/utilis.py:239-245).
COCO_ANCHORS_LARGEST_FIRST = [
    [116, 90], [156, 198], [373, 326],
    [30, 61], [62, 45], [59, 119],
    [10, 13], [16, 30], [33, 23],
]


def _conv(filters, size, stride, bn=True, act="leaky"):
    lines = ["[convolutional]"]
    if bn:
        lines.append("batch_normalize=1")
    lines += [f"filters={filters}", f"size={size}", f"stride={stride}",
              "pad=1", f"activation={act}"]
    return lines


def _res(a, b):
    """conv a/1x1, conv b/3x3, shortcut from=-3."""
    return [_conv(a, 1, 1), _conv(b, 3, 1),
            ["[shortcut]", "from=-3", "activation=linear"]]


def _head(a, b, det_filters):
    blocks = []
    for _ in range(3):
        blocks.append(_conv(a, 1, 1))
        blocks.append(_conv(b, 3, 1))
    blocks.append(_conv(det_filters, 1, 1, bn=False, act="linear"))
    return blocks


def yolov3_blocks_text(num_classes=80, num_anchors=3):
    det = (num_classes + 5) * num_anchors
    blocks = []
    # Darknet-53 backbone: blocks 0..74
    blocks.append(_conv(32, 3, 1))
    blocks.append(_conv(64, 3, 2))
    blocks += _res(32, 64)
    blocks.append(_conv(128, 3, 2))
    for _ in range(2):
        blocks += _res(64, 128)
    blocks.append(_conv(256, 3, 2))
    for _ in range(8):
        blocks += _res(128, 256)          # ends at block 36
    blocks.append(_conv(512, 3, 2))
    for _ in range(8):
        blocks += _res(256, 512)          # ends at block 61
    blocks.append(_conv(1024, 3, 2))
    for _ in range(4):
        blocks += _res(512, 1024)         # ends at block 74
    assert len(blocks) == 75
    yolo = ["[yolo]", f"classes={num_classes}", f"num={3 * num_anchors}"]
    # head 1 (stride 32): 75..82
    blocks += _head(512, 1024, det)
    blocks.append(list(yolo))
    # 83..94
    blocks.append(["[route]", "layers=-4"])
    blocks.append(_conv(256, 1, 1))
    blocks.append(["[upsample]", "stride=2"])
    blocks.append(["[route]", "layers=-1, 61"])
    blocks += _head(256, 512, det)
    blocks.append(list(yolo))
    # 95..106
    blocks.append(["[route]", "layers=-4"])
    blocks.append(_conv(128, 1, 1))
    blocks.append(["[upsample]", "stride=2"])
    blocks.append(["[route]", "layers=-1, 36"])
    blocks += _head(128, 256, det)
    blocks.append(list(yolo))
    assert len(blocks) == 107
    return blocks


def yolov3_cfg_text(num_classes=80, num_anchors=3):
    """Return the cfg file text (107 blocks, no [net])."""
    out = ["# YOLOv3 / Darknet-53 graph in the block layout yolo_v3.py expects",
           "# (no [net] block: hyper-parameters live in the params dict)", ""]
    for i, blk in enumerate(yolov3_blocks_text(num_classes, num_anchors)):
        out.append(f"# block {i}")
        out.extend(blk)
        out.append("")
    return "\n".join(out) + "\n"


def write_yolov3_cfg(path=None, num_classes=80, num_anchors=3):
    """Write the cfg to ``path`` (a fresh temp file when None); return the path."""
    text = yolov3_cfg_text(num_classes, num_anchors)
    if path is None:
        fd, path = tempfile.mkstemp(prefix="yolov3_", suffix=".cfg")
        os.close(fd)
    with open(path, "w") as fh:
        fh.write(text)
    return path


def default_params(height=416, num_classes=80, num_anchors=3, anchors=None,
                   cuda=False, confidence=0.5, nms_thesh=0.4, batch_size=1):
    """A params dict with every key the inference path reads
    (/root/reference/utilis.py:673-690, /root/reference/gghjgj.txt)."""
    import numpy as np
    if anchors is None:
        if num_anchors == 3:
            anchors = COCO_ANCHORS_LARGEST_FIRST
        else:
            # one representative anchor per head, largest first
            anchors = [COCO_ANCHORS_LARGEST_FIRST[3 * k + 1][:] for k in range(3)]
            anchors = [a for a in anchors for _ in range(num_anchors)][: 3 * num_anchors]
    return {
        "batch_size": batch_size,
        "height": height, "width": height, "channels": 3,
        "num_classes": num_classes, "num_anchors": num_anchors,
        "anchors": np.array(anchors, dtype=np.float64),
        "pretrain_snapshot": "", "cuda": cuda,
        "confidence": confidence, "specific_conf": confidence,
        "nms_thesh": nms_thesh, "iou_conf": 0.5,
        "classes": [f"class_{i}" for i in range(num_classes)],
        "conf_list": list(np.arange(start=0.2, stop=0.95, step=0.025)),
    }
