"""What the reference's callers do around the two hot-path calls, as functions.

``detection.py`` / ``video.py`` map the rows of ``filter_results`` from the
letterboxed network frame back to every original image and clamp them
(/root/reference/detection.py:52-70, /root/reference/video.py:78-91);
``evaluate.get_map`` runs ``filter_results`` once per image and per confidence of
a list of (by default) 30 values (/root/reference/evaluate.py:185-207,
/root/reference/utilis.py:688).  The functions here do the same work on the GPU:
the rescale as one HIP kernel over all rows, the confidence sweep with ONE NMS.
``DetectionPipeline`` is the callers' batch loop (detection.py:43-50,
video.py:60-66) with the NMS of one batch running on a second HIP stream beside
the forward pass of the next.
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


IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def preprocess_frames(frames, inp_dim, minmax=True, div255=False, mean=None, std=None,
                      return_canvas=False, device=None):
    """Raw frames to the network input on the GPU: what the data set of detection.py
    does per image on the host with LetterBoxImage_cv -> ImgToTensorCv -> Normalize
    (/root/reference/detection.py:31-35, /root/reference/utilis.py:69-116), or
    prep_image with minmax=False, div255=True (/root/reference/utilis.py:512-516).

    frames: sequence of (h, w, 3) uint8 BGR arrays as cv2.imread returns them, or
    uint8 torch tensors of that shape (device tensors are read where they are; host
    arrays are uploaded).  Returns (images, im_dim_list): images is the (B, 3,
    inp_dim, inp_dim) float32 device tensor model(images) takes, im_dim_list the
    [(width, height), ...] list the rescale step takes.  With return_canvas the
    letterboxed uint8 frames (B, inp_dim, inp_dim, 3) come as a third value (what
    LetterboxImage returns, /root/reference/utilis.py:89-104).

    The resize is OpenCV's 8-bit INTER_CUBIC arithmetic restated in the kernel; cv2
    itself was not available to compare against, so treat a one-level difference
    from cv2.resize in a small share of pixels as possible."""
    import ctypes
    _lib.require_gpu()
    if len(frames) == 0:
        raise ValueError("preprocess_frames: no frames")
    dev = torch.device(device if device is not None else "cuda")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    held, descs, dims = [], (_lib.ImageDesc * len(frames))(), []
    for i, f in enumerate(frames):
        t = f if isinstance(f, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(f))
        if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
            raise ValueError(f"frame {i}: expected (h, w, 3) uint8, got {tuple(t.shape)} {t.dtype}")
        if t.stride(2) != 1 or t.stride(1) != 3:
            t = t.contiguous()
        t = t.to(dev, non_blocking=True)
        held.append(t)
        descs[i].data = t.data_ptr()
        descs[i].height, descs[i].width = int(t.shape[0]), int(t.shape[1])
        descs[i].pitch_bytes = int(t.stride(0))
        dims.append((int(t.shape[1]), int(t.shape[0])))
    B, H = len(frames), int(inp_dim)
    images = torch.empty((B, 3, H, H), dtype=torch.float32, device=dev)
    canvas = torch.empty((B, H, H, 3), dtype=torch.uint8, device=dev) if return_canvas else None
    flags = (1 if minmax else 0) | (2 if div255 else 0)
    m = (ctypes.c_float * 3)(*mean) if mean is not None else None
    sd = (ctypes.c_float * 3)(*std) if std is not None else None
    nbytes = lib.yolo_amd_preprocess_workspace_bytes(B, H)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib.yolo_amd_preprocess(descs, B, H, flags, m, sd, images.data_ptr(),
                                      canvas.data_ptr() if canvas is not None else None,
                                      ws.data_ptr(), nbytes, _lib.current_stream_ptr(dev)),
              "preprocess")
    # the frames and the workspace are read by kernels queued on the current stream
    stream = torch.cuda.current_stream(dev)
    for t in held:
        t.record_stream(stream)
    ws.record_stream(stream)
    return (images, dims, canvas) if return_canvas else (images, dims)


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


_STREAM_PAIRS = {}


def concurrent_streams(device, n=2):
    """`n` HIP streams of `device` whose kernels really run side by side.  The runtime
    maps streams onto a few hardware queues (four by default), and two streams that land
    on the same queue run one after the other; which ones do is not exposed, so candidates
    are tried with a short spin kernel on each stream until `n` of them overlap (cached per
    device; if fewer can be found the last candidates tried fill the list)."""
    import time
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device(), int(n))
    if key in _STREAM_PAIRS:
        return _STREAM_PAIRS[key]
    with torch.cuda.device(device):
        spin_us = 500                                  # the library's idle-wave kernel, half a millisecond

        def wall(streams):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for st in streams:
                _lib.check(_lib.lib.yolo_amd_spin(spin_us, st.cuda_stream), "spin")
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        chosen = [torch.cuda.Stream(device=device)]
        wall(chosen)
        alone = min(wall(chosen) for _ in range(3))
        tries = 0
        while len(chosen) < n and tries < 16:
            tries += 1
            cand = torch.cuda.Stream(device=device)
            wall(chosen + [cand])
            together = min(wall(chosen + [cand]) for _ in range(3))
            if together < 1.5 * alone or tries == 16:
                chosen.append(cand)
        while len(chosen) < n:
            chosen.append(torch.cuda.Stream(device=device))
    _STREAM_PAIRS[key] = tuple(chosen)
    return _STREAM_PAIRS[key]


class DetectionPipeline:
    """The callers' loop ``for images in loader: prediction = model(images);
    output = filter_results(prediction, params)`` (detection.py:43-50) with two batches
    in flight: batch i runs its whole chain (forward pass, decode, NMS) on one HIP stream
    while batch i+1 runs on another, so the ramp and the drain of one batch's kernels are
    filled by the other's (about +11 % images/s at batch 32 on one MI355X over running
    them one after the other).  Results come back in submission order and are the
    tensors filter_results would have returned.

        pipe = DetectionPipeline(model, params)
        for images in loader:
            for output in pipe.submit(images):     # the outputs that are finished: none for the
                ...                                # first two batches, then one per call
        for output in pipe.flush():                # the last two
            ...

    Every output is a tensor of its own (nothing is recycled under the caller).  Batches
    may differ in shape (the last, partial batch of a loader).
    """

    LANES = 2

    def __init__(self, model, params, depth=None):
        """depth: batches in flight (default 2; a third adds about 2 % more images/s at
        batch 32 and a batch more of latency)."""
        _lib.require_gpu()
        self.model, self.params = model, params
        self.device = next(model.parameters()).device
        self.LANES = int(depth) if depth else DetectionPipeline.LANES
        self.lanes = list(concurrent_streams(self.device, self.LANES))
        self._n = 0
        self._pending = []              # PendingDetections of the batches in flight, oldest first

    def submit(self, images):
        """Queue forward + NMS for `images`.  Returns the list of outputs that this call
        retires: empty while fewer than `depth` batches are in flight, else the output of the
        oldest one (whose stream this batch is queued behind)."""
        from .utilis import filter_results_async
        main = torch.cuda.current_stream(self.device)
        st = self.lanes[self._n % self.LANES]
        self._n += 1
        done = [self._pending.pop(0).result()] if len(self._pending) >= self.LANES else []
        st.wait_stream(main)                            # the images were produced on the caller's stream
        if images.is_cuda:
            images.record_stream(st)
        with torch.cuda.device(self.device), torch.cuda.stream(st):
            # this stream's own plan, activation arena and NMS workspace
            self._pending.append(filter_results_async(self.model(images), self.params))
        return done

    def flush(self):
        """The outputs of the batches still in flight, oldest first (an empty list if none)."""
        out = []
        while self._pending:
            out.append(self._pending.pop(0).result())
        return out
