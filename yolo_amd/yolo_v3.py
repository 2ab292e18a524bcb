"""Drop-in for the reference's ``yolo_v3`` module: class ``yolo_v3`` with the
constructor, ``forward``, ``predict`` and ``load_weights`` of
/root/reference/yolo_v3.py, whose inference path runs on the gfx950 HIP
library instead of torch modules.

``forward(x)`` in eval() mode executes the fused plan (yolo_amd/engine.py): 75
implicit-GEMM conv launches (bf16 MFMA, BatchNorm folded, LeakyReLU, shortcut
add, upsample + route concat all fused) and one decode launch.  In train() mode
(the callers that never call eval(): /root/reference/video.py:21-25,
get_fps.py:25-29) the same convolution kernels run with unfolded weights and every
BatchNorm2d normalises on batch statistics in HIP kernels of its own, updating the
modules' running statistics in place as torch does.  The module
tree (``module_list`` with conv_<i> / bach_norm_<i> / leaky_<i>) is kept only
so that ``state_dict()`` / ``load_state_dict()`` / ``.cuda()`` behave like the
reference's; it is the source the plan folds its weights from.

Training (``fit``, ``yolo_loss``, ``forward(is_training=True)``) is outside
this package.
"""
from __future__ import annotations

import warnings

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import HeadDesc, check, lib
from .engine import Plan, UnsupportedGraph
from .utilis import create_module, box_iou, _save_checkpoint  # noqa: F401  (re-exported names)


def decode_heads(raws, anchors_per_head, height, num_classes):
    """Decode raw NCHW head tensors [(B, A*(5+C), G, G), ...] into one
    (B, sum G*G*A, 5+C) fp32 tensor with the HIP decode kernel
    (/root/reference/yolo_v3.py:205-262 and the torch.cat at :488-493)."""
    D = 5 + num_classes
    first = raws[0]
    src_device = first.device
    xs = []
    for r in raws:
        if not r.is_cuda:
            _lib.require_gpu()
            r = r.cuda()
        xs.append(r.float().contiguous())
    B = xs[0].shape[0]
    A = xs[0].shape[1] // D
    descs = (HeadDesc * len(xs))()
    rows = 0
    for k, (x, anc) in enumerate(zip(xs, anchors_per_head)):
        G = x.shape[2]
        stride = height // G
        d = descs[k]
        d.raw, d.G, d.layout, d.row_offset = x.data_ptr(), G, _lib.LAYOUT_NCHW, rows
        d.stride = float(stride)
        for a in range(A):
            # the division happens in float64, the result is rounded to fp32
            # (yolo_v3.py:248-250)
            d.anchor_w[a] = np.float32(anc[a][0] / stride)
            d.anchor_h[a] = np.float32(anc[a][1] / stride)
        rows += G * G * A
    out = torch.empty((B, rows, D), dtype=torch.float32, device=xs[0].device)
    with torch.cuda.device(out.device):
        check(lib.yolo_amd_decode(descs, len(xs), B, A, D, rows, out.data_ptr(),
                                  _lib.current_stream_ptr(out.device)), "decode")
    return out if src_device.type == "cuda" else out.to(src_device)


class yolo_v3(nn.Module):
    def __init__(self, params, blocks,
                 weight_file="../4TrainingWeights/yolov3.weights",
                 detection_layers=[81, 93, 105]):
        super().__init__()
        self.blocks = blocks
        self.params = params            # shared with the caller, mutated by callers
        for layer in detection_layers:
            self.blocks[layer]["filters"] = (self.params["num_classes"] + 5) * \
                self.params["num_anchors"]
        self.layer_type_dic, self.module_list = create_module(self.params, blocks)
        self.mse_loss = nn.MSELoss()
        self.bce_loss = nn.BCELoss()
        self.losses_name = ["total_loss", "x", "y", "w", "h", "conf", "cls"]
        self._plans = {}
        self._weights_version = 0
        self._device = None
        self._running_stats_moved = False   # a train()-mode forward has updated them since the last fold
        self._train_event = None            # recorded behind the last train()-mode forward pass
        if self.params["pretrain_snapshot"]:
            self.load_state_dict(torch.load(self.params["pretrain_snapshot"]))
        elif weight_file is not None:
            self.load_weights(weight_file, cust_train_zero=True)
        if self.params["cuda"]:
            self.cuda()

    # ------------------------------------------------------------ plan cache
    def _invalidate(self):
        self._plans = {}
        self._weights_version += 1
        self._device = None               # looked up again at the next forward

    def _param_device(self):
        if self._device is None:
            self._device = next(self.parameters()).device
        return self._device

    def load_state_dict(self, *args, **kwargs):
        res = super().load_state_dict(*args, **kwargs)
        self._invalidate()
        return res

    def _apply(self, fn, *args, **kwargs):      # .cuda() / .to() / .float() move the weights
        res = super()._apply(fn, *args, **kwargs)
        self._invalidate()
        return res

    def refold(self):
        """Call after modifying parameters in place (an optimizer step, manual
        edits): the fused plan folds BatchNorm into its own weight copies."""
        self._invalidate()

    def _plan(self, B, H, train=False):
        # One plan (activation arena, split-K scratch, descriptors) per shape AND per
        # stream: calls queued on one stream run in order and share it, calls on two
        # streams may overlap on the GPU and get arenas of their own.
        # A train()-mode plan is the exception: its kernels read and write the modules' own
        # running statistics, so there is ONE per shape whatever the stream, and forward()
        # runs such passes strictly one after the other (see _train_event).
        dev = self._param_device()
        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" and not train else 0
        key = (B, H, stream, train)
        plan = self._plans.get(key)
        if plan is None:
            plan = Plan(self, B, H, train=train)
            self._plans[key] = plan
        return plan

    # ------------------------------------------------------------ inference
    def forward(self, x, is_training=False, labels=None):
        if is_training:
            raise NotImplementedError("training (yolo_loss) is outside this package; use the "
                                      "reference for fit()")
        if x.dim() != 4 or x.shape[2] != x.shape[3]:
            raise ValueError("expected a (B, C, H, H) batch")
        src_device = x.device
        dev = self._param_device()
        if dev.type != "cuda":
            raise _lib.YoloAmdError("the model is on the CPU: call .cuda() (params['cuda']=True); "
                                    "the fused path has no CPU implementation")
        if x.device != dev:
            x = x.to(dev)
        if self.training:
            # video.py / get_fps.py never call eval(): BatchNorm then runs on batch
            # statistics and moves its running statistics with every call
            try:
                plan = self._plan(int(x.shape[0]), int(x.shape[2]), train=True)
            except UnsupportedGraph as exc:
                warnings.warn(f"yolo_v3 is in train() mode and the batch-statistics plan cannot take this "
                              f"model ({exc}); running the torch modules", stacklevel=2)
                out = self._forward_modules(x)
                return out if src_device == dev else out.to(src_device)
            # Batch-statistics passes update running_mean / running_var / num_batches_tracked
            # with read-modify-writes: two of them side by side on two streams (a pipeline
            # with several batches in flight over a model that never called eval()) would
            # lose updates.  Each pass waits for the one before it, on whatever stream that ran.
            cur = torch.cuda.current_stream(dev)
            if self._train_event is not None:
                cur.wait_event(self._train_event)
            self._running_stats_moved = True
            out = plan.run(x)
            ev = torch.cuda.Event()
            ev.record(cur)
            self._train_event = ev
            return out if src_device == dev else out.to(src_device)
        if self._running_stats_moved:
            # The eval() plans folded running statistics that have changed since.  The
            # passes that moved them, and eval() passes still reading the old folded
            # weights on other streams, must be over before the weights are replaced, and
            # the new ones in place before any stream uses them: a rare transition
            # (train() -> eval()), so the whole device is synchronised around it.
            torch.cuda.synchronize(dev)
            for key, plan in self._plans.items():
                if not key[3]:
                    plan.refold()
            torch.cuda.synchronize(dev)
            self._running_stats_moved = False
        out = self._plan(int(x.shape[0]), int(x.shape[2])).run(x)
        return out if src_device == dev else out.to(src_device)

    def _forward_modules(self, x):
        """Block-by-block execution with the torch modules
        (/root/reference/yolo_v3.py:449-496), decode through the HIP kernel."""
        cache = {}
        ltd = self.layer_type_dic
        raws, ancs = [], []
        for index, module in enumerate(self.module_list):
            kind = self.blocks[index]["type"]
            if kind in ("convolutional", "upsample"):
                x = module(x)
            elif kind == "route":
                ref = ltd["referred_relationship"][index]
                x = cache[ref] if not isinstance(ref, tuple) else \
                    torch.cat((cache[ref[0]], cache[ref[1]]), 1)
            elif kind == "shortcut":
                x += cache[ltd["referred_relationship"][index]]
            elif kind == "yolo":
                raws.append(x)
                ancs.append(np.asarray(module[0].anchors, dtype=np.float64))
                if len(raws) == 3:
                    return decode_heads(raws, ancs, self.params["height"],
                                        self.params["num_classes"])
            if index in ltd["referred"]:
                cache[index] = x

    def predict(self, x, layer, batch_size):
        """Decode one raw head (B, A*(5+C), G, G) -> (B, G*G*A, 5+C)
        (/root/reference/yolo_v3.py:205-262)."""
        anchors = np.asarray(layer[0].anchors, dtype=np.float64)
        return decode_heads([x], [anchors], self.params["height"], self.params["num_classes"])

    # ------------------------------------------------------------ weights
    def load_weights(self, weightfile, cust_train_zero=False):
        """Read a Darknet .weights file (/root/reference/yolo_v3.py:498-617):
        5 x int32 header, then per convolution in block order
        [bn.bias, bn.weight, running_mean, running_var] or [conv.bias], then the
        conv weights, all fp32.  With cust_train_zero the convolutions in front
        of the yolo blocks are zero-filled and the file pointer skips a
        constant 255 values twice, exactly as the reference does."""
        with open(weightfile, "rb") as fp:
            header = np.fromfile(fp, dtype=np.int32, count=5)
            weights = np.fromfile(fp, dtype=np.float32)
        self.header = torch.from_numpy(header)
        self.seen = self.header[3]
        zeroed = [i - 1 for i in self.layer_type_dic["yolo"]] if cust_train_zero else []
        ptr = 0

        def take(n):
            nonlocal ptr
            chunk = torch.from_numpy(weights[ptr:ptr + n])
            ptr += n
            return chunk

        with torch.no_grad():
            for i, block in enumerate(self.blocks):
                if block["type"] != "convolutional":
                    continue
                seq = self.module_list[i]
                conv = seq[0]
                try:
                    has_bn = int(block["batch_normalize"])
                except Exception:
                    has_bn = 0
                if has_bn:
                    bn = seq[1]
                    n = bn.bias.numel()
                    bn.bias.copy_(take(n).view_as(bn.bias))
                    bn.weight.copy_(take(n).view_as(bn.weight))
                    bn.running_mean.copy_(take(n).view_as(bn.running_mean))
                    bn.running_var.copy_(take(n).view_as(bn.running_var))
                elif i in zeroed:
                    conv.bias.zero_()
                    ptr += 255
                else:
                    conv.bias.copy_(take(conv.bias.numel()).view_as(conv.bias))
                if i in zeroed:
                    conv.weight.zero_()
                    ptr += 255
                else:
                    conv.weight.copy_(take(conv.weight.numel()).view_as(conv.weight))
        self._invalidate()


Darknet = yolo_v3
