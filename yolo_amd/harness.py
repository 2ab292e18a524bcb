"""What the reference's callers do around the two hot-path calls, as functions.

``detection.py`` / ``video.py`` map the rows of ``filter_results`` from the
letterboxed network frame back to every original image and clamp them
(/root/reference/detection.py:52-70, /root/reference/video.py:78-91);
``evaluate.get_map`` runs ``filter_results`` once per image and per confidence of
a list of (by default) 30 values (/root/reference/evaluate.py:185-207,
/root/reference/utilis.py:688).  The functions here do the same work on the GPU:
the rescale as one HIP kernel over all rows, the confidence sweep with ONE NMS.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from ._lib import check, lib
from .utilis import filter_results


def rescale_detections(output, im_dim_list, height):
    """detection.py:52-70.  output: the (K, 9) tensor of filter_results (or the int
    0, returned unchanged); im_dim_list: (B, 2) original (width, height) per image;
    height: the network input side.  Returns a new (K, 9) tensor with columns 1-4 in
    original-image pixels, clamped to the image."""
    if isinstance(output, int):
        return output
    _lib.require_gpu()
    dev_in = output.device
    rows = output.detach().to(device="cuda" if dev_in.type != "cuda" else dev_in, dtype=torch.float32).clone()
    rows = rows.contiguous()
    if rows.dim() != 2 or rows.shape[1] != 9:
        raise IndexError(f"output must be (K, 9), got {tuple(rows.shape)}")
    dims = torch.as_tensor(np.asarray(im_dim_list, dtype=np.float32).reshape(-1, 2)).to(rows.device).contiguous()
    with torch.cuda.device(rows.device):
        check(lib.yolo_amd_rescale_boxes(rows.data_ptr(), rows.shape[0], dims.data_ptr(), dims.shape[0],
                                         float(height), _lib.current_stream_ptr(rows.device)),
              "rescale_boxes")
    return rows if dev_in.type == "cuda" else rows.to(dev_in)


def filter_results_multi(prediction, params, conf_list):
    """filter_results(prediction, params) for every confidence of conf_list, as
    evaluate.get_map asks for it (evaluate.py:185-192), with one NMS instead of
    len(conf_list).

    Greedy NMS visits the boxes of a class in descending objectness and a box is
    only ever suppressed by a box visited before it, so the rows kept at a higher
    confidence threshold are exactly the rows kept at the lowest threshold whose
    objectness passes the higher one, in the same order.  Returns a list with one
    (K_i, 9) tensor, or the int 0, per confidence."""
    confs = [float(c) for c in conf_list]
    if not confs:
        return []
    low = dict(params)
    low["confidence"] = min(confs)
    base = filter_results(prediction, low)
    out = []
    for c in confs:
        if isinstance(base, int):
            out.append(0)
            continue
        # the candidate test of filter_results: objectness > conf (fp32 compare)
        sel = base[base[:, 5] > np.float32(c)]
        out.append(sel if sel.shape[0] else 0)
    return out


def detections_by_conf(prediction, params, conf_list):
    """The ``all_detections`` structure evaluate.get_map fills
    (evaluate.py:185-207): result[conf][image][class] = (n, 5) float array
    [x1, y1, x2, y2, objectness], rows in ascending objectness (ties keep the order
    of filter_results, where the reference's unstable argsort leaves them open)."""
    C = int(params["num_classes"])
    B = int(prediction.shape[0])
    per_conf = filter_results_multi(prediction, params, conf_list)
    result = []
    for rows in per_conf:
        images = [[np.array([]) for _ in range(C)] for _ in range(B)]
        if not isinstance(rows, int):
            r = rows.detach().cpu().numpy()
            for b in range(B):
                rb = r[r[:, 0] == b]
                if rb.shape[0] == 0:
                    continue
                rb = rb[np.argsort(rb[:, 5], kind="stable")]
                for c in range(C):
                    images[b][c] = rb[rb[:, 7] == c][:, 1:6]
        result.append(images)
    return result
