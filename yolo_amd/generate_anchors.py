"""Anchor k-means on label widths and heights, same surface as
/root/reference/generate_anchors.py (IOU :7-22, get_anchors :25-34, avg_IOU :37-44,
kmeans :64-96, generate_anchor :99-125, set_anchors_to_model :128-142).

Host-side numpy: it runs once per data set, before any model exists, on a few
thousand (w, h) pairs.  The per-sample Python loops of the reference are whole-array
expressions here; the accumulation order of the centroid sums is kept (np.add.at
adds in sample order), so the anchors come out bit-identical.
"""
from __future__ import annotations

import random
from os.path import join

import numpy as np


def IOU(x, centroids):
    """Overlap of a (w, h) box, or an (N, 2) array of them, with every centroid when
    all are anchored at one corner; (k,) or (N, k).  The four cases are tested in
    the reference's order, so a tie (equal width or height) takes the first formula
    that admits it (/root/reference/generate_anchors.py:7-22)."""
    x = np.asarray(x)
    c = np.asarray(centroids)
    w, h = x[..., 0, None], x[..., 1, None]
    cw, ch = c[:, 0], c[:, 1]
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.select(
            [(cw >= w) & (ch >= h), (cw >= w) & (ch <= h), (cw <= w) & (ch >= h)],
            [w * h / (cw * ch), w * ch / (w * h + (cw - w) * ch), cw * h / (w * h + cw * (ch - h))],
            (cw * ch) / (w * h))


def get_anchors(centroids, width_in_cfg_file, height_in_cfg_file):
    """Centroids scaled to cfg pixels: (anchors, order of descending width, anchors in
    that order) (/root/reference/generate_anchors.py:25-34)."""
    anchors = centroids * np.array([width_in_cfg_file, height_in_cfg_file])
    order = np.argsort(anchors[:, 0])[::-1]
    return anchors, order, anchors[order]


def avg_IOU(X, centroids):
    """Mean over samples of the best centroid overlap, summed in sample order
    (/root/reference/generate_anchors.py:37-44)."""
    total = 0.0
    for best in IOU(X, centroids).max(axis=1):
        total += best
    return total / X.shape[0]


def write_anchors_to_file(centroids, X, anchor_file, width_in_cfg_file, height_in_cfg_file):
    """Despite its name the reference writes nothing (the file output is commented
    out, generate_anchors.py:47-61): prints and returns the width-sorted anchors."""
    anchors_sort = get_anchors(centroids, width_in_cfg_file, height_in_cfg_file)[2]
    print("Anchors = ", anchors_sort)
    return anchors_sort


def kmeans(X, centroids, eps, anchor_file, width_in_cfg_file, height_in_cfg_file):
    """Lloyd iterations under the distance 1 - IOU until the assignment repeats;
    `centroids` is updated in place, `eps` is unused as in the reference
    (/root/reference/generate_anchors.py:64-96).  A centroid that loses every sample
    turns NaN in the reference, which then never terminates (every sample flips
    between two NaN centroids); here that raises ValueError instead."""
    k = centroids.shape[0]
    previous = np.full(X.shape[0], -1)
    while True:
        assignments = np.argmin(1 - IOU(X, centroids), axis=1)
        if (assignments == previous).all():
            return write_anchors_to_file(centroids, X, anchor_file,
                                         width_in_cfg_file, height_in_cfg_file)
        sums = np.zeros(centroids.shape, dtype=np.float64)
        np.add.at(sums, assignments, X)
        counts = np.bincount(assignments, minlength=k)
        if (counts == 0).any():
            raise ValueError(f"k-means: centroid(s) {np.where(counts == 0)[0].tolist()} lost "
                             "every sample; draw other initial centroids or use fewer clusters")
        centroids[:] = sums / counts[:, None]
        previous = assignments


def generate_anchor(label_csv, width_in_cfg_file, height_in_cfg_file, num_clusters=3):
    """Anchors from the label file's width/height columns (from the fifth column on),
    initial centroids drawn with random.randrange as the reference does; with
    num_clusters == 0 runs 1..10 clusters and returns the last
    (/root/reference/generate_anchors.py:99-125)."""
    import pandas as pd
    dims = np.array(pd.read_csv(label_csv).iloc[:, 4:].values, dtype=np.float64)
    anchors = None
    for k in (range(1, 11) if num_clusters == 0 else [num_clusters]):
        picks = [random.randrange(dims.shape[0]) for _ in range(k)]
        anchors = kmeans(dims, dims[picks], 0.005, join("anchors%d.txt" % k),
                         width_in_cfg_file, height_in_cfg_file)
    return anchors


def set_anchors_to_model(model, num_anchors, label_csv_mame, input_w, input_h):
    """k-means 3 x num_anchors anchors and hand them to the detection layers.  The
    reference walks the detection layers from the back, so the widest group goes to
    the LAST one in file order (/root/reference/generate_anchors.py:128-142); kept.
    Cached launch plans of the model hold the old anchors and are dropped."""
    anchors = generate_anchor(label_csv_mame, input_w, input_h, num_clusters=num_anchors * 3)
    for index, layer in enumerate(model.layer_type_dic["yolo"][::-1]):
        model.module_list[layer][0].anchors = anchors[index * num_anchors:(index + 1) * num_anchors]
    if hasattr(model, "refold"):
        model.refold()
