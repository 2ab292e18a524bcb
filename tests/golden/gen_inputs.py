"""Deterministic synthetic inputs shared by make_golden.py (which feeds them to
the reference) and by the tests (which feed the same arrays to the oracle and
to the HIP kernels).  Everything comes from numpy's legacy RandomState, whose
streams are frozen across numpy versions, so only expected OUTPUTS need to be
committed as fixtures.
"""
from __future__ import annotations

import numpy as np


def rs(seed):
    return np.random.RandomState(seed)


# ------------------------------------------------------------------ decode
def head_raw(seed, B, A, C, G, scale=1.5):
    """Raw head tensor (B, A*(5+C), G, G) float32, N(0, scale^2)."""
    return (rs(seed).standard_normal((B, A * (5 + C), G, G)) * scale).astype(np.float32)


# ------------------------------------------------------------------ IoU
def iou_pairs(seed, m=2048, H=416):
    """(m,4),(m,4) corner boxes incl. hand-made special cases at the front."""
    r = rs(seed)
    def boxes(k):
        cx, cy = r.uniform(0, H, k), r.uniform(0, H, k)
        w, h = r.uniform(1, 0.5 * H, k), r.uniform(1, 0.5 * H, k)
        return np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
    a, b = boxes(m), boxes(m)
    inf = np.inf
    special = [
        ([0, 0, 10, 10], [0, 0, 10, 10]),            # identical
        ([0, 0, 10, 10], [20, 20, 30, 30]),          # disjoint
        ([0, 0, 10, 10], [10, 0, 20, 10]),           # touching: +1 makes them overlap
        ([0, 0, 10, 10], [11, 0, 20, 10]),           # one pixel apart: iw = 0
        ([5, 5, 5, 5], [5, 5, 5, 5]),                # zero-area (area 1 with +1)
        ([0, 0, 10, 10], [2, 2, 4, 4]),              # contained
        ([-inf, 0, inf, 10], [0, 0, 10, 10]),        # exp overflow in w
        ([-inf, -inf, inf, inf], [-inf, -inf, inf, inf]),
        ([0, 0, -1, 10], [0, 0, 10, 10]),            # width + 1 == 0
        ([np.nan, 0, 10, 10], [0, 0, 10, 10]),       # NaN propagates
        ([0, 0, 10, 10], [0, 0, 10, np.nan]),
        ([1e30, 1e30, 3e38, 3e38], [1e30, 1e30, 3e38, 3e38]),  # area overflows
    ]
    for i, (p, q) in enumerate(special):
        a[i], b[i] = p, q
    return a.astype(np.float32), b.astype(np.float32)


# ------------------------------------------------------------------ NMS
def nms_pred(seed, B, N, C, H=416, mode="uniform", tie_levels=0,
             obj_lo=0.0, obj_hi=1.0):
    """Decoded prediction tensor (B, N, 5+C) float32.

    mode "uniform":   centres U(0,H), w,h U(8, 8+0.3H)
    mode "clustered": boxes jittered around 20 centres per image so that
                      suppression really happens
    Objectness is all-distinct per image (a permuted grid inside
    (obj_lo, obj_hi)) unless tie_levels > 0, in which case it is quantised to
    that many levels so equal-objectness groups far larger than 16 appear.
    """
    r = rs(seed)
    D = 5 + C
    pred = np.empty((B, N, D), dtype=np.float32)
    for b in range(B):
        if mode == "uniform":
            cx, cy = r.uniform(0, H, N), r.uniform(0, H, N)
            w, h = r.uniform(8, 8 + 0.3 * H, N), r.uniform(8, 8 + 0.3 * H, N)
        else:
            k = 20
            ccx, ccy = r.uniform(0.1 * H, 0.9 * H, k), r.uniform(0.1 * H, 0.9 * H, k)
            cw, ch = r.uniform(20, 0.4 * H, k), r.uniform(20, 0.4 * H, k)
            which = r.randint(0, k, N)
            cx = ccx[which] + r.normal(0, 4, N)
            cy = ccy[which] + r.normal(0, 4, N)
            w = cw[which] * np.exp(r.normal(0, 0.08, N))
            h = ch[which] * np.exp(r.normal(0, 0.08, N))
        grid = (r.permutation(N) + 0.5) / N
        obj = obj_lo + (obj_hi - obj_lo) * grid
        if tie_levels:
            obj = np.round(obj * tie_levels) / tie_levels
        cls = r.uniform(0, 1, (N, C))
        if mode == "clustered":
            # each cluster prefers one class so per-class groups are crowded
            fav = (which * 7) % C
            cls[np.arange(N), fav] += 0.6
            cls = np.clip(cls, 0, 1.6) / 1.6
        pred[b, :, 0], pred[b, :, 1] = cx, cy
        pred[b, :, 2], pred[b, :, 3] = w, h
        pred[b, :, 4] = obj
        pred[b, :, 5:] = cls
    return pred


# ------------------------------------------------------------------ weights
def synthetic_state_dict(table, seed=0):
    """Deterministic weights in the reference's state_dict key layout
    (module_list.<i>.conv_<i>.*, module_list.<i>.bach_norm_<i>.*) for the layer
    table produced from the cfg.  BatchNorm statistics are non-trivial so the
    BN fold is exercised; gains are chosen so activations stay O(1..10)
    through the 23 residual blocks and the head outputs have a spread that
    makes objectness straddle 0.5.
    Returns a dict of numpy float32 arrays (int64 for num_batches_tracked).
    """
    sd = {}
    for i, ent in enumerate(table):
        if ent["type"] != "conv":
            continue
        r = rs(seed * 1000 + i)
        cin, cout, k = ent["cin"], ent["cout"], ent["k"]
        fan_in = cin * k * k
        gain = 1.0
        nxt = table[i + 1]["type"] if i + 1 < len(table) else ""
        if nxt == "shortcut":
            gain = 0.2             # damp the residual branch
        if ent["stride"] == 2:
            gain = 0.8             # down-sampling convs
        if not ent["bn"]:
            gain = 0.3             # detection conv: raw head std ~1-2
        w = r.standard_normal((cout, cin, k, k)) * (gain * np.sqrt(2.0 / fan_in))
        sd[f"module_list.{i}.conv_{i}.weight"] = w.astype(np.float32)
        if ent["bn"]:
            p = f"module_list.{i}.bach_norm_{i}."
            sd[p + "weight"] = r.uniform(0.8, 1.2, cout).astype(np.float32)
            sd[p + "bias"] = r.normal(0, 0.1, cout).astype(np.float32)
            sd[p + "running_mean"] = r.normal(0, 0.1, cout).astype(np.float32)
            sd[p + "running_var"] = r.uniform(0.6, 1.4, cout).astype(np.float32)
            sd[p + "num_batches_tracked"] = np.array(0, dtype=np.int64)
        else:
            sd[f"module_list.{i}.conv_{i}.bias"] = r.normal(0, 0.5, cout).astype(np.float32)
    return sd


def images(seed, B, H):
    """Synthetic COCO-shaped batch: U(0,1) pixels then the ImageNet mean/std
    normalisation the callers apply (/root/reference/detection.py:34-35)."""
    x = rs(seed).uniform(0, 1, (B, 3, H, H)).astype(np.float32)
    mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(1, 3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], np.float32).reshape(1, 3, 1, 1)
    return ((x - mean) / std).astype(np.float32)


# ------------------------------------------------------------------ edits
# Hand-made modifications applied to nms_pred() outputs for the sentinel cases.
def _single_box(pred):
    pred = pred.copy()
    pred[:, :, 4] = 0.1
    pred[1, 17, 4] = 0.9
    return pred


def _middle_empty(pred):
    pred = pred.copy()
    pred[1, :, 4] *= 0.4
    return pred


def _overflow(pred):
    pred = pred.copy()
    # a high-objectness box whose width overflowed, one whose both sides did,
    # and a masked row carrying inf (inf * 0 = NaN in the reference's tensor)
    pred[0, 5, 2] = np.inf;  pred[0, 5, 4] = 0.99
    pred[0, 9, 2] = np.inf;  pred[0, 9, 3] = np.inf; pred[0, 9, 4] = 0.98
    pred[0, 11, 3] = np.inf; pred[0, 11, 4] = 0.05
    pred[1, 3, 2] = np.inf;  pred[1, 3, 4] = 0.31
    return pred


def _some_zero_obj(pred):
    pred = pred.copy()
    pred[0, ::3, 4] = 0.0
    return pred


EDITS = {"single_box": _single_box, "middle_empty": _middle_empty,
         "overflow": _overflow, "some_zero_obj": _some_zero_obj}


# ------------------------------------------------------------------ mAP scoring
def eval_case(seed, B, N, C, H=416, L=12, hit=0.7):
    """A data-set batch for the scoring code: labels (list of B arrays (L, 5)
    [class, cx, cy, w, h] in fractions of H, zero rows as padding, one image with no
    label at all when B > 2) and a decoded prediction (B, N, 5+C) whose first rows
    are jittered copies of the labelled boxes (a share `hit` of them with the right
    class), the rest clutter with low objectness."""
    r = rs(seed)
    pred = nms_pred(seed + 1000, B, N, C, H=H)
    pred[:, :, 4] *= 0.35
    labels = []
    for b in range(B):
        lab = np.zeros((L, 5), dtype=np.float64)
        n = 0 if (B > 2 and b == 1) else int(r.randint(1, L + 1))
        cls = r.randint(0, C, n)
        cxy = r.uniform(0.15, 0.85, (n, 2))
        wh = r.uniform(0.05, 0.3, (n, 2))
        lab[:n, 0] = cls
        lab[:n, 1:3] = cxy
        lab[:n, 3:5] = wh
        labels.append(lab)
        for k in range(n):
            for rep in range(2):          # two detections per object: one must lose
                row = pred[b, 2 * k + rep]
                jit = r.normal(0, 0.02 + 0.05 * rep, 4)
                row[0:2] = (cxy[k] + jit[:2]) * H
                row[2:4] = wh[k] * (1 + jit[2:]) * H
                row[4] = r.uniform(0.3, 1.0)
                row[5:] = r.uniform(0, 0.3, C)
                right = r.uniform() < hit
                row[5 + (cls[k] if right else r.randint(0, C))] = r.uniform(0.6, 1.0)
    return pred.astype(np.float32), labels


# ------------------------------------------------------------------ anchor k-means
def label_dims(seed, n, clusters=5):
    """(n, 2) label widths/heights as fractions of the image: a mixture of
    `clusters` log-normal blobs, plus a few exact duplicates and equal-side boxes so
    the tie branches of the overlap formula are taken."""
    r = rs(seed)
    centres = r.uniform(0.04, 0.6, (clusters, 2))
    which = r.randint(0, clusters, n)
    d = centres[which] * np.exp(r.normal(0, 0.25, (n, 2)))
    d = np.round(np.clip(d, 0.005, 0.99), 4)
    d[1] = d[0]
    d[3, 0] = d[2, 0]
    d[5, 1] = d[4, 1]
    return d


def write_label_csv(path, dims, seed=0):
    """A label file in the layout generate_anchor reads: width and height are the
    fifth and sixth columns (img_name, c, x, y, w, h)."""
    r = rs(seed)
    with open(path, "w") as fh:
        fh.write("img_name,c,x,y,w,h\n")
        for k, (w, h) in enumerate(dims):
            fh.write(f"img{k % 37}.jpg,{r.randint(0, 4)},{r.uniform():.4f},{r.uniform():.4f},{float(w)!r},{float(h)!r}\n")


# ------------------------------------------------------------------ image transforms
def photo(seed, w, h, lo=0, hi=256):
    """(h, w, 3) uint8 BGR test image: smooth gradients, a few hard-edged rectangles
    and noise, so cubic overshoot (clipping at 0 and 255) happens."""
    r = rs(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 120 * np.sin(xx / (7.0 + c) + yy / (11.0 + 2 * c)) for c in range(3)], 2)
    for _ in range(6):
        x0, y0 = r.randint(0, w), r.randint(0, h)
        x1, y1 = x0 + r.randint(1, max(2, w // 3)), y0 + r.randint(1, max(2, h // 3))
        img[y0:y1, x0:x1] = r.choice([0, 255], 3)
    img += r.normal(0, 6, img.shape)
    return np.clip(img, lo, hi - 1).astype(np.uint8)
