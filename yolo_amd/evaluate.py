"""Scoring side of the reference's evaluate.py: detections of the hot path in, mAP
out.  Same names and argument meaning as /root/reference/evaluate.py
(bbox_iou_numpy :14-49, compute_map :76-151, get_map :153-259).

The model call and the NMS inside get_map run through the HIP library
(yolo_amd.harness.detections_by_conf: one NMS pass at the lowest confidence, the
higher confidences are row filters of it).  The matching and the precision/recall
arithmetic are a few thousand scalar operations on host arrays and stay in numpy,
as they are in the reference.
"""
from __future__ import annotations

import numpy as np

from .utilis import compute_ap

_EPS = np.finfo(np.float64).eps


def bbox_iou_numpy(box1, box2):
    """(N, M) IoU of corner boxes, no +1 pixel convention, union floored at the
    float64 epsilon (/root/reference/evaluate.py:14-49)."""
    box1 = np.asarray(box1)
    box2 = np.asarray(box2)
    a = box1[:, None, :]
    b = box2[None, :, :]
    w = np.maximum(np.minimum(a[..., 2], b[..., 2]) - np.maximum(a[..., 0], b[..., 0]), 0)
    h = np.maximum(np.minimum(a[..., 3], b[..., 3]) - np.maximum(a[..., 1], b[..., 1]), 0)
    area_a = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
    area_b = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    union = np.maximum(area_a + area_b - w * h, _EPS)
    return (w * h) / union


def _match_image(detections, annotations, iou_conf):
    """True-positive flags of one image and class.  Detections claim annotations in
    stored order (ascending objectness, as get_map stores them): a detection is a
    true positive when its best-overlapping annotation reaches iou_conf and no
    earlier detection claimed it (/root/reference/evaluate.py:90-110)."""
    n = detections.shape[0] if detections.ndim == 2 else 0
    flags = np.zeros(n, dtype=np.int64)
    if n == 0 or annotations.shape[0] == 0:
        return flags
    overlaps = bbox_iou_numpy(detections[:, :4], annotations)
    best = np.argmax(overlaps, axis=1)
    claimed = np.zeros(annotations.shape[0], dtype=bool)
    for d in range(n):
        j = best[d]
        if overlaps[d, j] >= iou_conf and not claimed[j]:
            claimed[j] = True
            flags[d] = 1
    return flags


def class_average_precision(per_image_detections, per_image_annotations, iou_conf):
    """AP of one class over a data set; 0 when the class has no annotation
    (/root/reference/evaluate.py:112-141)."""
    flags, scores, num_annotations = [], [], 0
    for det, ann in zip(per_image_detections, per_image_annotations):
        num_annotations += ann.shape[0]
        if det.ndim == 2 and det.shape[0]:
            scores.append(det[:, 4])
            flags.append(_match_image(det, ann, iou_conf))
    if num_annotations == 0:
        return 0
    tp = np.concatenate(flags) if flags else np.zeros(0)
    sc = np.concatenate(scores) if scores else np.zeros(0)
    order = np.argsort(-sc)
    tp = tp[order]
    fp = np.cumsum(1 - tp)
    tp = np.cumsum(tp)
    recall = tp / num_annotations
    precision = tp / np.maximum(tp + fp, _EPS)
    return compute_ap(recall, precision)


def compute_map(all_detections, all_annotations, conf_index,
                map_frame, train, actual_num_labels, params):
    """mAP of all_detections[conf_index] against all_annotations
    (/root/reference/evaluate.py:76-151).  Returns (mAP, [AP per class]) when train,
    else (mAP, [AP per class], map_frame) with column params['confidence'] filled."""
    num_images = len(all_annotations)
    aps = []
    for label in range(params["num_classes"]):
        dets = [all_detections[conf_index][i][label] for i in range(num_images)]
        anns = [all_annotations[i][label] for i in range(num_images)]
        aps.append(class_average_precision(dets, anns, params["iou_conf"]))
    mAP = sum(aps) / actual_num_labels
    if train:
        return mAP, aps
    conf = params["confidence"]
    print(f"Average Precisions when confidence = {conf}:")
    for name, ap in zip(params["classes"], aps):
        map_frame.loc[name, conf] = ap
        print(f"+ Class '{name}' - AP: {ap}")
    print(f"mAP: {mAP}")
    map_frame.loc["mAP", conf] = mAP
    return mAP, aps, map_frame


def annotations_by_class(label, img_size, num_classes):
    """One image's label rows [class, cx, cy, w, h] (fractions of the input size;
    padding rows have w == h == 0 and a last column of 0) as per-class corner boxes
    in input pixels, plus the number of distinct classes, or None when the image has
    no label row (/root/reference/evaluate.py:211-229)."""
    label = np.asarray(label)
    per_class = [np.array([]) for _ in range(num_classes)]
    keep = label[:, -1] > 0
    if not keep.any():
        return per_class, None
    cls = label[keep, 0]
    xywh = label[keep, 1:]
    boxes = np.empty_like(xywh)
    boxes[:, 0] = xywh[:, 0] - xywh[:, 2] / 2
    boxes[:, 1] = xywh[:, 1] - xywh[:, 3] / 2
    boxes[:, 2] = xywh[:, 0] + xywh[:, 2] / 2
    boxes[:, 3] = xywh[:, 1] + xywh[:, 3] / 2
    boxes *= img_size
    for c in range(num_classes):
        per_class[c] = boxes[cls == c, :]
    return per_class, len(np.unique(cls))


def get_map(model, dataloader, train=False, loop_conf=False, confidence=False):
    """mAP of a model over a data loader at one or several object confidences
    (/root/reference/evaluate.py:153-259); same return tuple.

    The loader yields {"image": (B, C, H, W), "label": sequence of (L, 5) arrays}.
    actual_num_labels follows the reference: the largest value seen of "distinct
    classes in the most recent image that had a label" (the reference leaves it
    unbound, and fails, when the very first image has none; here it starts at 0).
    """
    import torch
    from .harness import detections_by_conf

    params = model.params
    if confidence:
        loop_conf = confidence
    elif loop_conf:
        loop_conf = params["conf_list"]
    else:
        loop_conf = [params["specific_conf"]]
    map_frame = None
    if not train:
        import pandas as pd
        map_frame = pd.DataFrame(index=params["classes"] + ["mAP"], columns=loop_conf)
    if hasattr(model, "eval"):
        model.eval()
    num_classes = params["num_classes"]
    all_detections = [[] for _ in loop_conf]
    all_annotations = []
    actual_num_labels = 0
    num_labels = 0
    with torch.no_grad():
        for samples in dataloader:
            image, labels = samples["image"], samples["label"]
            if params["cuda"]:
                image = image.to("cuda")
            img_size = image.shape[-1]
            outputs = model(image)
            batch = detections_by_conf(outputs, params, loop_conf)
            for ci in range(len(loop_conf)):
                all_detections[ci].extend(batch[ci])
            for label in labels:
                per_class, n = annotations_by_class(label, img_size, num_classes)
                all_annotations.append(per_class)
                if n is not None:
                    num_labels = n
                actual_num_labels = max(actual_num_labels, num_labels)
    best_map = best_ap = best_conf = None
    specific_conf_map = specific_conf_ap = None
    for ci, conf in enumerate(loop_conf):
        params["confidence"] = conf
        print(f"Running for object confidence : {conf}")
        print(f"actual_num_labels : {actual_num_labels}")
        results = compute_map(all_detections, all_annotations, ci, map_frame, train,
                              actual_num_labels, params)
        if ci == 0 or results[0] > best_map:
            best_map, best_ap, best_conf = results[0], results[1], conf
        if np.round(conf, 3) == params["specific_conf"]:
            specific_conf_map, specific_conf_ap = results[0], results[1]
    if hasattr(model, "train"):
        model.train()
    if train:
        return best_map, best_ap, best_conf, specific_conf_map, specific_conf_ap
    return best_map, best_ap, best_conf, specific_conf_map, specific_conf_ap, map_frame
