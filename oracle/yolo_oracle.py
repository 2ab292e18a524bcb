"""CPU oracle for the YOLOv3 inference hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, on the CPU, what the reference computes on the path
``model(images)`` -> ``filter_results(prediction, params)``.  It exists so the
HIP kernels can be checked against something that follows the reference's
arithmetic operation by operation.  Nothing in the product package may import
it: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` do, and only as the checker.

Pinned against the reference itself: ``tests/golden/make_golden.py`` imports
/root/reference (utilis.py, yolo_v3.py) in the build container, runs it on
seeded inputs and commits the input/output vectors under ``tests/golden/``;
``tests/test_oracle_golden.py`` replays them through this file.

Reference lines restated here:
  parse_cfg            /root/reference/utilis.py:126-166
  create_module        /root/reference/utilis.py:169-295
  yolo_v3.forward      /root/reference/yolo_v3.py:449-496
  yolo_v3.predict      /root/reference/yolo_v3.py:205-262
  box_iou              /root/reference/utilis.py:298-323
  filter_img_class     /root/reference/utilis.py:355-390
  filter_results       /root/reference/utilis.py:393-465
  rescale block        /root/reference/detection.py:52-70 (= video.py:78-91)
  image transforms     /root/reference/utilis.py:69-116, 495-516 and the Normalize
                       of /root/reference/detection.py:31-35.  The pad / min-max /
                       channel flip / mean-std part is pinned (golden_pre.npz holds
                       what numpy and torch return for the reference's statements).
                       The cubic resize is cv2.resize, which is not installed here
                       and which the reference holds no output of: resize_cubic_u8
                       restates OpenCV's published 8-bit fixed-point algorithm and
                       its parity with cv2 is UNPINNED.
"""
from __future__ import annotations

import numpy as np


# --------------------------------------------------------------------------
# cfg text -> blocks -> layer table
# --------------------------------------------------------------------------
def parse_cfg_text(text):
    """utilis.py:126-166.  Values stay strings; blank and '#' lines dropped."""
    blocks, cur = [], {}
    for raw in text.split("\n"):
        if len(raw) == 0 or raw[0] == "#":
            continue
        line = raw.strip()
        if line[0] == "[":
            if cur:
                blocks.append(cur)
                cur = {}
            cur["type"] = line[1:-1].strip()
        else:
            key, value = line.split("=")
            cur[key.rstrip()] = value.lstrip()
    blocks.append(cur)
    return blocks


def parse_cfg(path):
    with open(path, "r") as fh:
        return parse_cfg_text(fh.read())


def layer_table(blocks, channels, num_classes, num_anchors,
                detection_layers=(81, 93, 105)):
    """Walk the blocks the way create_module does (utilis.py:248-295) and
    return one dict per block:
      conv:     type, cin, cout, k, stride, pad, bn(bool), leaky(bool)
      upsample: type, stride
      route:    type, srcs (absolute block indices, 1 or 2)
      shortcut: type, src (absolute)
      yolo:     type, head (0,1,2 in file order)
    plus ``cout`` (channels leaving the block) on every entry.
    """
    det = (num_classes + 5) * num_anchors
    table, out_ch = [], []
    cin = channels
    n_yolo = 0
    for i, blk in enumerate(blocks):
        t = blk["type"]
        if t == "convolutional":
            cout = det if i in detection_layers else int(blk["filters"])
            k = int(blk["size"])
            ent = dict(type="conv", cin=cin, cout=cout, k=k,
                       stride=int(blk["stride"]), pad=0 if k == 1 else 1,
                       bn=("batch_normalize" in blk),
                       leaky=(blk["activation"] == "leaky"))
            cin = cout
        elif t == "upsample":
            ent = dict(type="upsample", stride=int(blk["stride"]))
        elif t == "route":
            items = [int(s) for s in blk["layers"].split(",")]
            first = i + items[0]
            if len(items) == 1:
                ent = dict(type="route", srcs=[first])
                cin = out_ch[first]
            elif len(items) == 2:
                # second item is used as an ABSOLUTE index (utilis.py:215-218)
                ent = dict(type="route", srcs=[first, items[1]])
                cin = out_ch[first] + out_ch[items[1]]
            else:
                raise Exception("route with more than two layers")
        elif t == "shortcut":
            ent = dict(type="shortcut", src=i + int(blk["from"]))
        elif t == "yolo":
            ent = dict(type="yolo", head=n_yolo)
            n_yolo += 1
        else:
            raise Exception(f"unknown block type {t!r}")
        ent["cout"] = cin
        out_ch.append(cin)
        table.append(ent)
    return table


# --------------------------------------------------------------------------
# forward (torch CPU fp32 ops)
# --------------------------------------------------------------------------
def _round_bf16(t):
    """fp32 -> nearest bf16 (ties to even) -> fp32: what storing an activation or a
    weight as bf16 does to it."""
    import torch
    return t.to(torch.bfloat16).to(torch.float32)


def forward(state_dict, blocks, params, x, return_layers=False, training=False, emulate_bf16=False,
            sum_dtype=None):
    """yolo_v3.forward with is_training=False (yolo_v3.py:449-496).  ``state_dict``
    uses the reference's key names (module_list.<i>.conv_<i>.weight,
    module_list.<i>.bach_norm_<i>.*).  ``x`` is a (B, C, H, H) float32 torch tensor
    on the CPU.

    training=False is the model after eval(): BatchNorm2d on its running statistics.
    training=True is the model as constructed (video.py:21-25 and get_fps.py:25-29
    never call eval()): every BatchNorm2d normalises with the batch's own mean and
    biased variance and moves running_mean / running_var (momentum 0.1, unbiased
    variance) and num_batches_tracked; the moved statistics are written back into
    ``state_dict`` (new tensors: the caller's originals are not modified in place).

    emulate_bf16=True (eval() only) is the same network with the storage roundings of
    the bf16 device path put in, and nothing else changed: BatchNorm folded into the
    convolution (scale = gamma / sqrt(var + eps) into the weights, the rest into an fp32
    bias), folded weights rounded to bf16 except those of the layer that reads the fp32
    image, sums accumulated in fp32, and every layer's output rounded ONCE to bf16 after
    bias, LeakyReLU and the shortcut add that follows it; the convolutions in front of
    the yolo blocks keep their fp32 sums, which are decoded in fp32.  It answers one
    question for the tests: how far is a bf16 kernel from what bf16 storage alone gives.
    The reference has no such mode (it is an fp32 program); this is a derived checker.
    sum_dtype (emulate_bf16 only; default float32): the type the convolution sums are
    accumulated in before the result is taken to float32 -- torch.float64 gives the same
    network with nothing changed but the rounding of the sums, which is how the tests show
    what a different summation ORDER alone does to a chain of bf16 roundings."""
    import torch
    import torch.nn.functional as F

    if emulate_bf16 and training:
        raise ValueError("emulate_bf16 restates the folded (eval) path only")
    C, A = params["num_classes"], params["num_anchors"]
    table = layer_table(blocks, params["channels"], C, A)
    anchors = np.asarray(params["anchors"], dtype=np.float64)
    outs = [None] * len(table)
    dets = []
    x = x.to(torch.float32)
    reads_image = True                 # the next convolution reads the network input
    with torch.no_grad():
        for i, ent in enumerate(table):
            t = ent["type"]
            nxt = table[i + 1]["type"] if i + 1 < len(table) else None
            if t == "conv" and emulate_bf16:
                w = state_dict[f"module_list.{i}.conv_{i}.weight"].to(torch.float32)
                b = state_dict.get(f"module_list.{i}.conv_{i}.bias")
                if ent["bn"]:
                    p = f"module_list.{i}.bach_norm_{i}."
                    scale = state_dict[p + "weight"].float() / torch.sqrt(state_dict[p + "running_var"].float() + 1e-5)
                    bias = state_dict[p + "bias"].float() - state_dict[p + "running_mean"].float() * scale
                    if b is not None:
                        bias = bias + b.float() * scale
                    w = w * scale.view(-1, 1, 1, 1)
                else:
                    bias = b.float() if b is not None else torch.zeros(w.shape[0])
                if not reads_image:
                    w = _round_bf16(w)
                if sum_dtype is None or sum_dtype == torch.float32:
                    x = F.conv2d(x, w, bias, stride=ent["stride"], padding=ent["pad"])
                else:
                    x = F.conv2d(x.to(sum_dtype), w.to(sum_dtype), bias.to(sum_dtype), stride=ent["stride"],
                                 padding=ent["pad"]).to(torch.float32)
                if ent["leaky"]:
                    x = F.leaky_relu(x, 0.1)
                if nxt not in ("shortcut", "yolo"):
                    x = _round_bf16(x)
                reads_image = False
            elif t == "conv":
                w = state_dict[f"module_list.{i}.conv_{i}.weight"]
                b = state_dict.get(f"module_list.{i}.conv_{i}.bias")
                x = F.conv2d(x, w, b, stride=ent["stride"], padding=ent["pad"])
                if ent["bn"]:
                    p = f"module_list.{i}.bach_norm_{i}."
                    if training:
                        rm = state_dict[p + "running_mean"].clone()
                        rv = state_dict[p + "running_var"].clone()
                        x = F.batch_norm(x, rm, rv, state_dict[p + "weight"], state_dict[p + "bias"],
                                         training=True, momentum=0.1, eps=1e-5)
                        state_dict[p + "running_mean"], state_dict[p + "running_var"] = rm, rv
                        key = p + "num_batches_tracked"
                        state_dict[key] = state_dict.get(key, torch.zeros((), dtype=torch.int64)) + 1
                    else:
                        x = F.batch_norm(x, state_dict[p + "running_mean"],
                                         state_dict[p + "running_var"],
                                         state_dict[p + "weight"],
                                         state_dict[p + "bias"],
                                         training=False, momentum=0.1, eps=1e-5)
                if ent["leaky"]:
                    x = F.leaky_relu(x, 0.1)
            elif t == "upsample":
                x = F.interpolate(x, scale_factor=ent["stride"], mode="nearest")
            elif t == "route":
                if len(ent["srcs"]) == 1:
                    x = outs[ent["srcs"][0]]
                else:
                    x = torch.cat((outs[ent["srcs"][0]], outs[ent["srcs"][1]]), 1)
            elif t == "shortcut":
                x = x + outs[ent["src"]]
                if emulate_bf16:
                    x = _round_bf16(x)
            elif t == "yolo":
                k = ent["head"]
                head_anchors = anchors[k * A:(k + 1) * A]
                raw = x
                x = decode_head(raw, head_anchors, params["height"], C)
                dets.append(x)
                if len(dets) == 3:
                    outs[i] = x
                    det = torch.cat(dets, 1)
                    return (det, outs) if return_layers else det
            outs[i] = x
    raise Exception("cfg has fewer than three yolo blocks")


def decode_head(raw, head_anchors, height, num_classes):
    """yolo_v3.predict (yolo_v3.py:205-262) in closed form.

    raw: (B, A*D, G, G) float32 torch tensor.  Returns (B, G*G*A, D) with
    row = (gy*G + gx)*A + a.  fp32 throughout; the anchor / stride division is
    done in float64 first and then rounded to float32 (yolo_v3.py:248-250).
    """
    import torch
    B, ch, G, _ = raw.shape
    D = 5 + num_classes
    A = ch // D
    stride = height // G
    x = raw.reshape(B, A, D, G, G).permute(0, 3, 4, 1, 2).contiguous()  # B,gy,gx,A,D
    x = x.to(torch.float32).clone()
    gx = torch.arange(G, dtype=torch.float32).view(1, 1, G, 1)
    gy = torch.arange(G, dtype=torch.float32).view(1, G, 1, 1)
    aw = torch.tensor([np.float32(a[0] / stride) for a in head_anchors],
                      dtype=torch.float32).view(1, 1, 1, A)
    ah = torch.tensor([np.float32(a[1] / stride) for a in head_anchors],
                      dtype=torch.float32).view(1, 1, 1, A)
    s = float(stride)
    out = torch.empty_like(x)
    out[..., 0] = (torch.sigmoid(x[..., 0]) + gx) * s
    out[..., 1] = (torch.sigmoid(x[..., 1]) + gy) * s
    out[..., 2] = (torch.exp(x[..., 2]) * aw) * s
    out[..., 3] = (torch.exp(x[..., 3]) * ah) * s
    out[..., 4] = torch.sigmoid(x[..., 4])
    out[..., 5:] = torch.sigmoid(x[..., 5:])
    return out.view(B, G * G * A, D)


# --------------------------------------------------------------------------
# IoU and NMS (numpy float32, operation order of the reference)
# --------------------------------------------------------------------------
def _nanmax(a, b):
    # torch.max(a, b) returns NaN when either operand is NaN; np.maximum too.
    return np.maximum(a, b)


def box_iou(box1, box2):
    """utilis.py:298-323.  +1 pixel convention, float32, no fused ops.
    box1: (1,4) or (m,4); box2: (m,4).  Returns float32 (m,)."""
    b1 = np.asarray(box1, dtype=np.float32)
    b2 = np.asarray(box2, dtype=np.float32)
    one = np.float32(1.0)
    zero = np.float32(0.0)
    with np.errstate(all="ignore"):
        ix1 = np.maximum(b1[:, 0], b2[:, 0])
        iy1 = np.maximum(b1[:, 1], b2[:, 1])
        ix2 = np.minimum(b1[:, 2], b2[:, 2])
        iy2 = np.minimum(b1[:, 3], b2[:, 3])
        iw = np.maximum((ix2 - ix1) + one, zero)      # clamp(min=0) keeps NaN
        ih = np.maximum((iy2 - iy1) + one, zero)
        inter = iw * ih
        a1 = ((b1[:, 2] - b1[:, 0]) + one) * ((b1[:, 3] - b1[:, 1]) + one)
        a2 = ((b2[:, 2] - b2[:, 0]) + one) * ((b2[:, 3] - b2[:, 1]) + one)
        union = (a1 + a2) - inter
        return (inter / union).astype(np.float32)


def _first_max(scores):
    """torch.max(x, 1) on (n, C): value and FIRST index of the maximum; a NaN
    beats every number and the first NaN wins."""
    n, C = scores.shape
    best = scores[:, 0].copy()
    idx = np.zeros(n, dtype=np.int64)
    for j in range(1, C):
        v = scores[:, j]
        take = (v > best) | (np.isnan(v) & ~np.isnan(best))
        best = np.where(take, v, best)
        idx = np.where(take, j, idx)
    return best, idx


def candidates(pred, conf, num_classes):
    """Steps (1)-(6) of filter_results (utilis.py:394-436) for one image.

    pred: (N, D) float32.  Returns (rows8, n_index): rows8 is (m, 8) float32
    [x1,y1,x2,y2,obj,cls_score,cls_idx,ymin] for the rows whose objectness is
    > conf (and > 0), in original order; n_index their original row numbers.
    """
    pred = np.asarray(pred, dtype=np.float32)
    obj = pred[:, 4]
    with np.errstate(all="ignore"):
        keep = (obj > np.float32(conf)) & (obj > np.float32(0))
    n_index = np.nonzero(keep)[0]
    p = pred[n_index]
    half = np.float32(2.0)
    with np.errstate(all="ignore"):
        x1 = p[:, 0] - p[:, 2] / half
        y1 = p[:, 1] - p[:, 3] / half
        x2 = p[:, 0] + p[:, 2] / half
        y2 = p[:, 1] + p[:, 3] / half
    if len(n_index):
        score, cidx = _first_max(p[:, 5:5 + num_classes])
    else:
        score = np.zeros(0, np.float32)
        cidx = np.zeros(0, np.int64)
    rows = np.stack([x1, y1, x2, y2, p[:, 4], score,
                     cidx.astype(np.float32), y1], axis=1).astype(np.float32)
    return rows, n_index


def greedy_nms_sorted(boxes, thresh):
    """Greedy suppression over boxes already in visiting order
    (utilis.py:367-385): a box is kept iff ``iou < thresh`` holds against
    every earlier KEPT box, so iou >= thresh and NaN both suppress.
    Returns the kept positions."""
    n = boxes.shape[0]
    alive = np.ones(n, dtype=bool)
    thresh = np.float32(thresh)
    kept = []
    for i in range(n):
        if not alive[i]:
            continue
        kept.append(i)
        if i + 1 < n:
            iou = box_iou(boxes[i:i + 1, :4], boxes[i + 1:, :4])
            with np.errstate(all="ignore"):
                ok = iou < thresh
            alive[i + 1:] &= ok
    return np.asarray(kept, dtype=np.int64)


def filter_results(pred, conf, nms_thresh, num_classes):
    """utilis.py:393-465 with the tie rule fixed to "objectness descending,
    then original row index ascending" (a stable sort; the reference's
    torch.sort is unstable for groups of 17 or more equal keys).

    pred: (B, N, D) float32 array.  Returns a (K, 9) float32 array
    [image, x1, y1, x2, y2, obj, cls_score, cls_idx, ymin], ordered image
    ascending, class ascending, objectness descending; or the int 0 when no
    row survives the confidence filter (utilis.py:462-465).
    """
    pred = np.asarray(pred, dtype=np.float32)
    out = []
    for b in range(pred.shape[0]):
        rows, n_index = candidates(pred[b], conf, num_classes)
        if rows.shape[0] == 0:
            continue
        for cls in np.unique(rows[:, 6]):
            sel = np.nonzero(rows[:, 6] == cls)[0]
            grp = rows[sel]
            # descending objectness, ties by ascending original row
            order = np.lexsort((n_index[sel], -grp[:, 4].astype(np.float64)))
            grp = grp[order]
            kept = greedy_nms_sorted(grp, nms_thresh)
            grp = grp[kept]
            img = np.full((grp.shape[0], 1), b, dtype=np.float32)
            out.append(np.concatenate([img, grp], axis=1))
    if not out:
        return 0
    return np.concatenate(out, axis=0).astype(np.float32)


# --------------------------------------------------------------------------
# The reference-shaped Python loop, used only for the CPU baseline timing:
# same per-box torch calls as utilis.py:355-465 so that its cost profile is
# the reference's.  Output equals filter_results() above on tie-free input.
# --------------------------------------------------------------------------
def filter_results_torch_loop(prediction, conf, nms_thresh, num_classes):
    import torch

    def iou_t(b1, b2):
        ix1 = torch.max(b1[:, 0], b2[:, 0])
        iy1 = torch.max(b1[:, 1], b2[:, 1])
        ix2 = torch.min(b1[:, 2], b2[:, 2])
        iy2 = torch.min(b1[:, 3], b2[:, 3])
        inter = torch.clamp(ix2 - ix1 + 1, min=0) * torch.clamp(iy2 - iy1 + 1, min=0)
        a1 = (b1[:, 2] - b1[:, 0] + 1) * (b1[:, 3] - b1[:, 1] + 1)
        a2 = (b2[:, 2] - b2[:, 0] + 1) * (b2[:, 3] - b2[:, 1] + 1)
        return inter / (a1 + a2 - inter)

    p = prediction * (prediction[:, :, 4] > conf).float().unsqueeze(2)
    corners = torch.empty_like(p[:, :, :4])
    corners[:, :, :2] = p[:, :, :2] - p[:, :, 2:4] / 2
    corners[:, :, 2:] = p[:, :, :2] + p[:, :, 2:4] / 2
    p[:, :, :4] = corners
    pieces = []
    for b in range(p.size(0)):
        img = p[b]
        img = img[img[:, 4] > 0]
        if img.shape[0] == 0:
            continue
        score, cidx = torch.max(img[:, 5:5 + num_classes], 1)
        img = torch.cat((img[:, :5], score.unsqueeze(1), cidx.float().unsqueeze(1),
                         img[:, 1:2]), 1)
        for cls in torch.unique(img[:, 6]):
            g = img[img[:, 6] == cls]
            g = g[torch.sort(g[:, 4], descending=True, stable=True)[1]]
            i = 0
            while i < g.size(0):
                if i + 1 < g.size(0):
                    m = (iou_t(g[i, :4].unsqueeze(0), g[i + 1:, :4]) < nms_thresh)
                    g = torch.cat((g[:i + 1], g[i + 1:][m]), 0)
                i += 1
            pieces.append(torch.cat((g.new_full((g.size(0), 1), b), g), 1))
    if not pieces:
        return 0
    return torch.cat(pieces, 0)


# --------------------------------------------------------------------------
# caller post-processing
# --------------------------------------------------------------------------
def rescale_boxes(rows, im_dims, height):
    """detection.py:52-70: filter_results rows (K, 9) from the letterboxed
    height x height frame back to each original image (im_dims: (B, 2) = width,
    height), clamped to it.  fp32 operation by operation; returns a new array.
    The block is inline in detection() (and video.py), so it is pinned by
    tests/golden/golden_post.npz, made by torch executing the same statements."""
    out = np.array(rows, dtype=np.float32, copy=True)
    if out.shape[0] == 0:
        return out
    dims = np.asarray(im_dims, dtype=np.float32)[out[:, 0].astype(np.int64)]     # :53-54
    hgt = np.float32(height)
    with np.errstate(all="ignore"):
        # :55-56  `int / tensor` is Tensor.__rtruediv__ = tensor.reciprocal() * int: two roundings
        rcp = (np.float32(1) / dims).astype(np.float32)
        sf = np.minimum(rcp[:, 0] * hgt, rcp[:, 1] * hgt).astype(np.float32)
        px = ((hgt - sf * dims[:, 0]) / np.float32(2)).astype(np.float32)        # :62-63
        py = ((hgt - sf * dims[:, 1]) / np.float32(2)).astype(np.float32)        # :64-65
        for col, pad, lim in ((1, px, dims[:, 0]), (3, px, dims[:, 0]), (2, py, dims[:, 1]), (4, py, dims[:, 1])):
            v = ((out[:, col] - pad) / sf).astype(np.float32)                    # :66
            v = np.where(np.isnan(v), v, np.minimum(np.maximum(v, np.float32(0)), lim))   # :67-71, NaN stays NaN
            out[:, col] = v
    return out


def filter_results_multi(pred, conf_list, nms_thresh, num_classes):
    """What evaluate.get_map does per batch (evaluate.py:185-192): filter_results
    once per confidence of the list.  Returns [rows or 0 per confidence]."""
    return [filter_results(pred, float(c), nms_thresh, num_classes) for c in conf_list]


# --------------------------------------------------------------------------
# image transforms of the callers
# --------------------------------------------------------------------------
def letterbox_geometry(img_w, img_h, w, h):
    """Size and offset of the resized image inside the w x h canvas, in Python
    doubles as the reference computes them (utilis.py:75-81)."""
    new_w = int(img_w * min(w / img_w, h / img_h))
    new_h = int(img_h * min(w / img_w, h / img_h))
    return new_w, new_h, (w - new_w) // 2, (h - new_h) // 2


def _cubic_taps(n_dst, n_src):
    """Per destination coordinate: first-tap source index minus nothing (the taps
    are src[s-1 .. s+2], clamped to the image) and the four 11-bit fixed-point
    weights.  OpenCV resize, INTER_CUBIC, 8-bit: position (d + 0.5) * scale - 0.5
    in double, rounded to float; Keys kernel with A = -0.75 in float; weights
    scaled by 2048 and rounded half-to-even to int16.  PARITY WITH cv2 UNPINNED."""
    scale = 1.0 / (n_dst / n_src)
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    x = (f - s.astype(np.float32)).astype(np.float32)
    A, one = np.float32(-0.75), np.float32(1.0)
    xp = x + one
    c0 = ((A * xp - np.float32(5) * A) * xp + np.float32(8) * A) * xp - np.float32(4) * A
    c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
    om = one - x
    c2 = ((A + np.float32(2)) * om - (A + np.float32(3))) * om * om + one
    c3 = one - c0 - c1 - c2
    wts = np.stack([c0, c1, c2, c3], 1).astype(np.float32) * np.float32(2048)
    wts = np.clip(np.rint(wts), -32768, 32767).astype(np.int64)
    return s, wts


def resize_cubic_u8(img, new_w, new_h):
    """(h, w, c) uint8 -> (new_h, new_w, c) uint8, bicubic without antialiasing:
    horizontal pass into integers, vertical pass, one rounding shift by 22 bits,
    saturation to 0..255; taps outside the image repeat the border pixel."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w, _ = img.shape
    sx, ax = _cubic_taps(new_w, w)
    sy, ay = _cubic_taps(new_h, h)
    src = img.astype(np.int64)
    rows = np.zeros((h, new_w, img.shape[2]), dtype=np.int64)
    for k in range(4):
        rows += src[:, np.clip(sx + k - 1, 0, w - 1), :] * ax[None, :, k, None]
    out = np.zeros((new_h, new_w, img.shape[2]), dtype=np.int64)
    for k in range(4):
        out += rows[np.clip(sy + k - 1, 0, h - 1)] * ay[:, k, None, None]
    return np.clip((out + (1 << 21)) >> 22, 0, 255).astype(np.uint8)


def letterbox_canvas(resized, w, h):
    """utilis.py:80-82 / 497-507: integer canvas of 128 with the resized image in
    the middle (the canvas is int64 in the reference; values fit a byte)."""
    new_h, new_w = resized.shape[:2]
    canvas = np.full((h, w, resized.shape[2]), 128)
    top, left = (h - new_h) // 2, (w - new_w) // 2
    canvas[top:top + new_h, left:left + new_w, :] = resized
    return canvas


def canvas_to_input(canvas, minmax=True, div255=False, mean=None, std=None):
    """Canvas (h, w, 3) BGR integers -> (3, h, w) float32 RGB network input.
    minmax: LetterBoxImage_cv's float64 (c - min) / (max - min) (utilis.py:85), then
    the float32 conversion of ImgToTensorCv (:107-116).  div255: prep_image's
    float32 division (:512-516).  mean/std: torchvision Normalize as detection.py
    :34-35 applies it, float32 subtract then float32 divide per channel."""
    canvas = np.asarray(canvas).astype(np.int64)
    if minmax:
        with np.errstate(divide="ignore", invalid="ignore"):
            val = (canvas - canvas.min()) / (canvas.max() - canvas.min())
    else:
        val = canvas
    t = np.ascontiguousarray(val[:, :, ::-1].transpose(2, 0, 1)).astype(np.float32)
    if div255:
        t = t / np.float32(255.0)
    if mean is not None:
        m = np.asarray(mean, dtype=np.float32)[:, None, None]
        s = np.asarray(std, dtype=np.float32)[:, None, None]
        t = (t - m) / s
    return t.astype(np.float32)


def preprocess(images, inp_dim, minmax=True, div255=False, mean=None, std=None):
    """The detection data pipeline for a list of (h, w, 3) uint8 BGR images:
    returns ((B, 3, H, H) float32, [canvases], (B, 2) original (w, h))."""
    outs, canvases, dims = [], [], []
    for img in images:
        ih, iw = img.shape[:2]
        new_w, new_h, _, _ = letterbox_geometry(iw, ih, inp_dim, inp_dim)
        canvas = letterbox_canvas(resize_cubic_u8(img, new_w, new_h), inp_dim, inp_dim)
        canvases.append(canvas.astype(np.uint8))
        outs.append(canvas_to_input(canvas, minmax, div255, mean, std))
        dims.append((iw, ih))
    return np.stack(outs), canvases, np.array(dims, dtype=np.float32)
