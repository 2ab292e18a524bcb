"""Drop-in for the reference's ``utilis`` module, backed by the gfx950 HIP
library (libyolo_amd.so) for everything on the inference hot path.

Same names, argument meaning and error behaviour as /root/reference/utilis.py
for what detection.py, video.py, get_fps.py and evaluate.py import:

  hot path (HIP kernels):  box_iou, filter_img_class, filter_results
  graph construction:      parse_cfg, create_module, empty_layer, yolo_layer
  small host helpers:      load_classes, compute_ap, my_collate,
                           my_collate_detection, worker_init_fn,
                           ImgToTensorCv, _save_checkpoint, prep_params,
                           prep_labels
  cv2-dependent helpers:   LetterBoxImage_cv, LetterboxImage, letterbox_image,
                           prep_image, write, detection_write (import cv2 lazily)

Aliases for the names the tutorial lineage uses: bbox_iou, write_results,
predict_transform; and LetterboxImage_cv, which get_fps.py imports although the
reference module only defines LetterBoxImage_cv.

There is no CPU implementation here: CPU tensors are moved to the GPU, run
through the HIP kernels and moved back.  Without a GPU the hot-path functions
raise.
"""
from __future__ import annotations

import ctypes
import datetime
import json
import logging
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import lib, check


# --------------------------------------------------------------------------
# placeholder modules (/root/reference/utilis.py:22-30)
# --------------------------------------------------------------------------
class empty_layer(nn.Module):
    def __init__(self):
        super().__init__()


class yolo_layer(nn.Module):
    def __init__(self, anchors):
        super().__init__()
        self.anchors = anchors


# --------------------------------------------------------------------------
# cfg -> blocks -> modules (/root/reference/utilis.py:126-295)
# --------------------------------------------------------------------------
def parse_cfg(cfgfile):
    """Text cfg -> list of dict blocks.  Blank lines and lines starting with
    '#' are dropped, '[type]' opens a block, 'key=value' pairs are stored with
    the spaces around '=' removed; every value stays a string."""
    with open(cfgfile, "r") as fh:
        text = fh.read()
    blocks = []
    block = {}
    for raw in text.split("\n"):
        if not raw or raw[0] == "#":
            continue
        line = raw.strip()
        if line[0] == "[":
            if block:
                blocks.append(block)
            block = {"type": line[1:-1].strip()}
        else:
            key, value = line.split("=")
            block[key.rstrip()] = value.lstrip()
    blocks.append(block)
    return blocks


def _add_conv(module, index, block, in_channel, registry):
    # bias is dropped whenever the key batch_normalize is present, whatever
    # its value (utilis.py:171-177); padding ignores the cfg's `pad`
    has_bn = "batch_normalize" in block
    filters = int(block["filters"])
    ksize = int(block["size"])
    module.add_module(f"conv_{index}",
                      nn.Conv2d(in_channel, filters, ksize, int(block["stride"]),
                                0 if ksize == 1 else 1, bias=not has_bn))
    if has_bn and block["batch_normalize"]:
        module.add_module(f"bach_norm_{index}", nn.BatchNorm2d(filters))   # (sic)
    if block["activation"] == "leaky":
        module.add_module(f"leaky_{index}", nn.LeakyReLU(0.1))
    registry["conv"].append(index)
    return filters


def _add_route(module, index, block, channels_so_far, registry):
    items = [int(v) for v in block["layers"].split(",")]
    first = index + items[0]                       # relative
    out = channels_so_far[first]
    if len(items) == 1:
        registry["referred_relationship"][index] = first
        registry["route_1"].append(index)
    elif len(items) == 2:
        second = items[1]                          # absolute
        out += channels_so_far[second]
        registry["referred"].append(second)
        registry["referred_relationship"][index] = (first, second)
        registry["route_2"].append(index)
    else:
        raise Exception("Route layer is not behaving as we planned, please "
                        "change the code according.")
    module.add_module(f"route_{index}", empty_layer())
    registry["referred"].append(first)
    return out


def create_module(params, blocks):
    """Build the nn.ModuleList (one nn.Sequential per block, sub-module names
    conv_<i> / bach_norm_<i> / leaky_<i> / upsample_<i> / route_<i> /
    short_cut_<i> / yolo_<i>) and the routing table ``layer_type_dic``.
    Side effect kept from the reference: params['width'] = params['height']."""
    module_list = nn.ModuleList()
    registry = {"conv": [], "upsampling": [], "shortcut": [], "route_1": [],
                "route_2": [], "yolo": [], "referred": [],
                "referred_relationship": {}}
    params["width"] = params["height"]
    in_channel = params["channels"]
    channels_so_far = []
    for index, block in enumerate(blocks):
        module = nn.Sequential()
        kind = block["type"]
        if kind == "convolutional":
            in_channel = _add_conv(module, index, block, in_channel, registry)
        elif kind == "upsample":
            module.add_module(f"upsample_{index}",
                              nn.Upsample(scale_factor=int(block["stride"]), mode="nearest"))
            registry["upsampling"].append(index)
        elif kind == "route":
            in_channel = _add_route(module, index, block, channels_so_far, registry)
        elif kind == "shortcut":
            src = index + int(block["from"])
            module.add_module(f"short_cut_{index}", empty_layer())
            registry["shortcut"].append(index)
            registry["referred"].append(src)
            registry["referred_relationship"][index] = src
        elif kind == "yolo":
            # the k-th yolo block in file order takes anchors[k*A:(k+1)*A]; the
            # cfg's own mask/anchors entries are ignored (utilis.py:239-245)
            first = len(registry["yolo"]) * params["num_anchors"]
            block["mask"] = range(first, first + params["num_anchors"])
            module.add_module(f"yolo_{index}", yolo_layer(params["anchors"][block["mask"]]))
            registry["yolo"].append(index)
        channels_so_far.append(in_channel)
        module_list.append(module)
    return registry, module_list


# --------------------------------------------------------------------------
# device plumbing
# --------------------------------------------------------------------------
def _to_device_f32(t):
    """(tensor on the GPU as contiguous fp32, original device)."""
    if not torch.is_tensor(t):
        raise TypeError("expected a torch tensor")
    src_device = t.device
    if not t.is_cuda:
        _lib.require_gpu()
        t = t.cuda()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous(), src_device


_WORKSPACES = {}


def _workspace(device, nbytes):
    """Scratch for the NMS kernels, one buffer per (device, stream): calls queued on
    one stream run in order and may share it, calls on different streams may overlap
    and must not.  A buffer that is outgrown is replaced; the caching allocator keeps
    the old block intact until the kernels already queued on its stream have run."""
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (index, torch.cuda.current_stream(device).cuda_stream)
    buf = _WORKSPACES.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _WORKSPACES[key] = buf
    return buf


# --------------------------------------------------------------------------
# hot path
# --------------------------------------------------------------------------
def box_iou(box1, box2):
    """IoU with the +1 pixel convention (/root/reference/utilis.py:298-323).
    Both arguments are 2-d (n, 4) corner boxes; box1 may hold one box that is
    broadcast.  Returns an (m,) fp32 tensor on box2's device."""
    b2, dev = _to_device_f32(box2)
    b1, _ = _to_device_f32(box1.to(b2.device) if box1.device != b2.device else box1)
    if b1.dim() != 2 or b2.dim() != 2:
        raise IndexError("too many indices for tensor of dimension 1")
    n1, m = b1.shape[0], b2.shape[0]
    if n1 != 1 and n1 != m:
        if m == 1:                      # broadcast the other way round
            b1, b2, n1, m = b2, b1, 1, n1
        else:
            raise RuntimeError(f"The size of tensor a ({n1}) must match the size of "
                               f"tensor b ({m}) at non-singleton dimension 0")
    out = torch.empty(m, dtype=torch.float32, device=b2.device)
    with torch.cuda.device(b2.device):
        check(lib.yolo_amd_box_iou(b1.data_ptr(), n1, b2.data_ptr(), m, out.data_ptr(),
                                   _lib.current_stream_ptr(b2.device)), "box_iou")
    return out if dev.type == "cuda" else out.to(dev)


def filter_img_class(cls, image_pred, ind, nms_thesh):
    """Per-class greedy NMS (/root/reference/utilis.py:355-390).
    image_pred: (m, 8) rows [x1,y1,x2,y2,obj,score,cls,ymin].  Returns the
    tuple (batch_ind (k,1), rows (k,8)) like the reference."""
    rows, dev = _to_device_f32(image_pred)
    m = rows.shape[0]
    if m == 0:
        empty = rows.new_zeros((0, 8))
        res = (rows.new_zeros((0, 1)), empty)
        return res if dev.type == "cuda" else tuple(r.to(dev) for r in res)
    out = torch.empty((m, 8), dtype=torch.float32, device=rows.device)
    cnt = torch.empty(1, dtype=torch.int32, device=rows.device)
    nbytes = lib.yolo_amd_filter_img_class_workspace_bytes(m)
    ws = _workspace(rows.device, nbytes)
    with torch.cuda.device(rows.device):
        check(lib.yolo_amd_filter_img_class(rows.data_ptr(), m, float(cls), float(nms_thesh),
                                            ws.data_ptr(), ws.numel(), out.data_ptr(),
                                            cnt.data_ptr(), _lib.current_stream_ptr(rows.device)),
              "filter_img_class")
    k = int(cnt.item())
    kept = out[:k]
    batch_ind = kept.new_full((k, 1), float(ind))
    res = (batch_ind, kept)
    return res if dev.type == "cuda" else tuple(r.to(dev) for r in res)


class _HostCount:
    """A pinned host int32 the scan kernel writes the row count K into directly (pinned
    memory is mapped into the device's address space), so that no copy kernel and no
    stream synchronisation stand between the NMS kernels and the host knowing K.
    The host arms it with -1 and polls; counters are recycled (pinning is slow).

    Every launch records `event` on the launch stream behind the LAST kernel (the gather
    that writes the rows runs after the scan that publishes K).  Two things hang on it:
    a consumer on another stream waits for it before touching the rows, and a counter
    only goes back to the pool once it has completed -- a counter dropped while its
    kernels are still queued is parked in `_limbo` until then, so that the late store
    of K can never land in memory that has been handed to somebody else."""
    _pool = []
    _limbo = []

    def __init__(self):
        self.tensor = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.view = self.tensor.numpy()
        self.event = torch.cuda.Event()
        self.stream = None
        self.recorded = False

    @classmethod
    def take(cls):
        if cls._limbo:
            still = []
            for hc in cls._limbo:
                (cls._pool if hc.event.query() else still).append(hc)
            cls._limbo[:] = still
        hc = cls._pool.pop() if cls._pool else cls()
        hc.view[0] = -1
        hc.stream = None
        return hc

    def launched(self, stream, record=True):
        """Call right after the kernels have been queued on `stream`.  record=False leaves the
        event out (recording one costs the stream a few microseconds): for the synchronous
        caller, who consumes the rows on the launch stream and hands the counter back only
        after K has arrived, i.e. after the one kernel that writes to it has run."""
        self.stream = stream
        self.recorded = bool(record)
        if record:
            self.event.record(stream)

    def give_back(self):
        """Return the counter: to the pool if the kernel that stores into it has run (K has
        arrived, or everything queued with it is over), else to the list of counters whose
        kernels are still in flight."""
        if self.stream is None or self.view[0] != -1:
            _HostCount._pool.append(self)
            return
        if not self.recorded:
            self.event.record(self.stream)      # a failed wait: park it behind what is queued now
            self.recorded = True
        if self.event.query():
            _HostCount._pool.append(self)
        else:
            _HostCount._limbo.append(self)

    def ready(self):
        return int(self.view[0]) != -1

    def wait(self):
        """Spin until the kernel has stored K (a few microseconds after the scan kernel
        ends).  Every few dozen polls the stream is queried as well: once everything
        queued has run, a synchronisation makes the store visible whatever the memory's
        coherence mode -- and surfaces a failed kernel instead of spinning for ever."""
        view = self.view
        stream = self.stream                 # the stream the kernels were queued on
        while True:
            for _ in range(256):
                if view[0] != -1:
                    return int(view[0])
            if stream.query():
                break
        stream.synchronize()
        if view[0] == -1:
            raise _lib.YoloAmdError("filter_results: the row count never reached the host")
        return int(view[0])

    def order_rows_behind(self, rows):
        """Make the rows safe to use on the caller's current stream.  K reaches the host
        before the gather kernel has written the rows; on the launch stream that is
        plain stream order, on any other stream the consumer has to wait for the launch
        stream's event (a device-side wait), and the allocator must not recycle the
        block under reads queued on that other stream."""
        cur = torch.cuda.current_stream(rows.device)
        if self.stream is not None and cur != self.stream:
            if not self.recorded:
                self.event.record(self.stream)
                self.recorded = True
            cur.wait_event(self.event)
            rows.record_stream(cur)


def _launch_filter_results(prediction, params, record_event=True):
    """Queue the NMS kernels on the current stream.  Returns (rows buffer, host
    count, source device), or None for an empty input."""
    pred, dev = _to_device_f32(prediction)
    if pred.dim() != 3:
        raise IndexError(f"prediction must be (B, N, 5+C), got {tuple(pred.shape)}")
    B, N, D = pred.shape
    C = int(params["num_classes"])
    if B == 0 or N == 0:
        return None
    out = torch.empty((B * N, 9), dtype=torch.float32, device=pred.device)
    nbytes = lib.yolo_amd_filter_results_workspace_bytes(B, N, C)
    ws = _workspace(pred.device, nbytes)
    cnt = _HostCount.take()
    stream = torch.cuda.current_stream(pred.device)
    try:
        with torch.cuda.device(pred.device):
            check(lib.yolo_amd_filter_results(pred.data_ptr(), B, N, D, C,
                                              float(params["confidence"]), float(params["nms_thesh"]),
                                              ws.data_ptr(), ws.numel(), out.data_ptr(), cnt.tensor.data_ptr(),
                                              _lib.current_stream_ptr(pred.device)),
                  "filter_results")
    finally:
        # also after a failed launch: some kernels may have been queued already
        cnt.launched(stream, record_event)
    return out, cnt, dev


def filter_results(prediction, params):
    """Confidence filter + per-(image, class) NMS
    (/root/reference/utilis.py:393-465).

    prediction: (B, N, 5+C) decoded boxes; never modified.  Returns a (K, 9)
    fp32 tensor [image, x1, y1, x2, y2, objectness, class score, class index,
    ymin] ordered image asc / class asc / objectness desc (ties by original
    row), or the Python int 0 when nothing passes the confidence threshold.

    The row count K is the one thing the host has to wait for (callers index
    output.shape[0]); the rows themselves are ready in stream order, like any
    other tensor a kernel has just written.
    """
    launched = _launch_filter_results(prediction, params, record_event=False)
    if launched is None:
        return 0
    out, cnt, dev = launched
    try:
        k = cnt.wait()
        cnt.order_rows_behind(out)
    finally:
        cnt.give_back()
    if k == 0:
        return 0
    res = out[:k]
    return res if dev.type == "cuda" else res.to(dev)


class PendingDetections:
    """What filter_results_async returns: the NMS is queued, nothing has been waited
    for.  result() blocks until the row count has reached the host and returns what
    filter_results would have returned."""

    def __init__(self, rows, host_count, dev):
        self._rows, self._host, self._dev = rows, host_count, dev

    def done(self):
        return self._host is None or self._host.ready()

    def result(self):
        """The rows, ready for use on the caller's CURRENT stream (which may differ from
        the stream the NMS was queued on: the rows are then ordered behind the launch
        stream's last kernel by a device-side event wait)."""
        if self._rows is None:
            return 0
        host, self._host = self._host, None
        try:
            k = host.wait()
            host.order_rows_behind(self._rows)
        finally:
            host.give_back()
        res = self._rows[:k] if k else 0
        self._rows = None
        if isinstance(res, int) or self._dev.type == "cuda":
            return res
        return res.to(self._dev)

    def __del__(self):
        # dropped without result(): the counter stays out of circulation until the
        # kernels that write to it have run
        host = getattr(self, "_host", None)
        if host is not None:
            self._host = None
            try:
                host.give_back()
            except Exception:      # interpreter shutdown
                pass


def filter_results_async(prediction, params):
    """filter_results without the wait: the kernels are queued on the current stream
    and a PendingDetections is returned at once, so a caller can queue the next batch's
    forward pass (or another stream's NMS) before it asks for result().  No counterpart
    in the reference, whose filter_results is synchronous."""
    launched = _launch_filter_results(prediction, params)
    if launched is None:
        return PendingDetections(None, None, None)
    out, cnt, dev = launched
    return PendingDetections(out, cnt, dev)


def predict_transform(prediction, inp_dim, anchors, num_classes, CUDA=True):
    """Tutorial-lineage name for the head decode: (B, A*(5+C), G, G) raw head ->
    (B, G*G*A, 5+C) decoded, anchors given in pixels."""
    from .yolo_v3 import decode_heads
    return decode_heads([prediction], [np.asarray(anchors, dtype=np.float64)], int(inp_dim),
                        int(num_classes))


bbox_iou = box_iou
write_results = filter_results


# --------------------------------------------------------------------------
# small host helpers the callers import
# --------------------------------------------------------------------------
def load_classes(namesfile):
    with open(namesfile, "r") as fh:
        return fh.read().split("\n")[:-1]


def compute_ap(recall, precision):
    """VOC-style area under the precision envelope
    (/root/reference/utilis.py:468-492)."""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([0.0], precision, [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]
    step = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[step + 1] - mrec[step]) * mpre[step + 1])


def my_collate(batch):
    return {"image": torch.stack([b["image"] for b in batch], 0),
            "label": [b["label"] for b in batch]}


def my_collate_detection(batch):
    return ([b["ori_image"] for b in batch],
            torch.stack([b["image"] for b in batch], 0),
            [b["img_name"] for b in batch],
            [b["dim"] for b in batch],
            [b["labels"] for b in batch])


def worker_init_fn(worker_id):
    np.random.seed(worker_id)


class ImgToTensorCv:
    """HWC BGR array -> CHW RGB float tensor (/root/reference/utilis.py:107-116)."""

    def __call__(self, img):
        return torch.FloatTensor(np.ascontiguousarray(img[:, :, ::-1].transpose(2, 0, 1)))


def _cv2():
    try:
        import cv2
    except ImportError as exc:      # pragma: no cover - cv2 is not in this image
        raise ImportError("this helper needs OpenCV (cv2), as the reference does") from exc
    return cv2


def _letterbox(img, w, h, channels):
    cv2 = _cv2()
    img_w, img_h = img.shape[1], img.shape[0]
    scale = min(w / img_w, h / img_h)
    new_w, new_h = int(img_w * scale), int(img_h * scale)
    resized = cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_CUBIC)
    canvas = np.full((h, w, channels), 128)
    top, left = (h - new_h) // 2, (w - new_w) // 2
    canvas[top:top + new_h, left:left + new_w, :] = resized
    return canvas


class LetterBoxImage_cv:
    """Aspect-preserving cubic resize onto a 128-grey canvas, then min-max
    normalisation to [0, 1] (/root/reference/utilis.py:69-86)."""

    def __init__(self, inp_dim, inp_channel=3):
        self.w, self.h = inp_dim
        self.inp_channel = inp_channel

    def __call__(self, img):
        canvas = _letterbox(img, self.w, self.h, self.inp_channel)
        return (canvas - canvas.min()) / (canvas.max() - canvas.min())


LetterboxImage_cv = LetterBoxImage_cv      # the spelling get_fps.py imports


class LetterboxImage:
    """Same canvas without the normalisation; accepts PIL images
    (/root/reference/utilis.py:89-104)."""

    def __init__(self, inp_dim, inp_channel=3):
        self.w, self.h = inp_dim
        self.inp_channel = inp_channel

    def __call__(self, img):
        return _letterbox(np.array(img), self.w, self.h, self.inp_channel)


def letterbox_image(img, inp_dim, inp_channel=3):
    return _letterbox(img, inp_dim[0], inp_dim[1], inp_channel)


def prep_image(img, inp_dim):
    img = letterbox_image(img, (inp_dim, inp_dim))
    img = np.ascontiguousarray(img[:, :, ::-1].transpose(2, 0, 1))
    return torch.from_numpy(img).float().div(255.0).unsqueeze(0)


def _draw(x, img, classes):
    cv2 = _cv2()
    c1 = tuple(x[1:3].int())
    c2 = tuple(x[3:5].int())
    label = "{0}".format(classes[int(x[7])])
    color = 2
    cv2.rectangle(img, c1, c2, color, 1)
    t_size = cv2.getTextSize(label, cv2.FONT_HERSHEY_PLAIN, 1, 1)[0]
    c2 = c1[0] + t_size[0] + 3, c1[1] + t_size[1] + 4
    cv2.rectangle(img, c1, c2, color, -1)
    cv2.putText(img, label, (c1[0], c1[1] + t_size[1] + 4), cv2.FONT_HERSHEY_PLAIN, 1,
                [225, 255, 255], 1)
    return img


def write(x, results, classes):
    """Draw one output row (cols 1-4 box, col 7 class) on a single frame."""
    return _draw(x, results, classes)


def detection_write(x, results, classes):
    """Draw one output row on the image its column 0 names."""
    return _draw(x, results[int(x[0])], classes)


def _save_checkpoint(model, save_txt=True):
    """state_dict -> <sub_working_dir>/<timestamp>.pth (+ params JSON .txt) and
    record the path in params['pretrain_snapshot']
    (/root/reference/utilis.py:655-670)."""
    stamp = str(datetime.datetime.now()).replace(" ", "_").replace(":", "_")
    path = os.path.join(model.params["sub_working_dir"], stamp + ".pth")
    model.params["pretrain_snapshot"] = path
    torch.save(model.state_dict(), path)
    logging.info(f"Model checkpoint saved to {path}")
    if save_txt:
        dump = dict(model.params)
        dump["anchors"] = np.asarray(dump["anchors"]).tolist()
        with open(os.path.join(model.params["sub_working_dir"], stamp + ".txt"), "w") as fh:
            fh.write(json.dumps(dump, indent=4))


def prep_labels(img_txt_path, name_list, label_csv_mame, selected_cls=False):
    """Collect Darknet label files into one csv (/root/reference/utilis.py:519-535):
    every line of every file matching the glob ``img_txt_path`` becomes a row whose
    first column is the file's path with 'txt' replaced by 'jpg' (every occurrence,
    as str.replace does) and whose other columns are the line's whitespace-separated
    items, kept as strings; ``selected_cls`` keeps the rows whose column ``c`` is in
    the list."""
    import glob
    import pandas as pd
    rows = []
    for path in glob.glob(img_txt_path):
        with open(path, "r") as fh:
            for line in fh:
                items = line.split()
                rows.append([path.replace("txt", "jpg")] + [items[i - 1] for i in range(1, len(name_list))])
    out_file = pd.DataFrame(rows, columns=name_list)
    if selected_cls:
        out_file = out_file[out_file.c.isin(selected_cls)]
    out_file.to_csv(label_csv_mame, index=False)


def prep_params(params_dir, label_csv_mame, experiment_params=False,
                names_file="../4Others/human_eye.names"):
    """Load the params JSON, merge overrides, turn the anchor list into an
    array -- or, when the file holds no list, run the anchor k-means over the
    label csv with num_anchors * 3 clusters -- and add 'classes' and
    'conf_list' (/root/reference/utilis.py:673-690)."""
    with open(params_dir) as fh:
        params = json.load(fh)
    if experiment_params:
        params = {**params, **experiment_params}
    params["width"] = params["height"]
    if type(params["anchors"]) == list:
        params["anchors"] = np.array(params["anchors"])
    else:
        from .generate_anchors import generate_anchor
        params["anchors"] = generate_anchor(label_csv_mame, params["width"], params["height"],
                                            num_clusters=params["num_anchors"] * 3)
    params["classes"] = load_classes(names_file)
    params["conf_list"] = list(np.arange(start=0.2, stop=0.95, step=0.025))
    print(params)
    return params
