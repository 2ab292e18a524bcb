"""Execution plan for the fused inference path.

Turns the cfg blocks + the model's parameters into what the HIP library runs:

  * BatchNorm folded into bf16 weights (K-tile major, see pack_conv_weights) and an fp32 bias
    (recomputed whenever weights are reloaded);
  * one ``yolo_amd_conv_desc`` per convolution, with the following shortcut
    block folded in as a residual operand and the preceding upsample / route
    blocks folded into the source addressing, so the 107-block graph
    (/root/reference/yolo_v3.py:449-496) becomes 75 kernel launches;
  * NHWC bf16 activation buffers placed in one arena by liveness, so the hot
    working set stays small enough to live in the 256 MiB Infinity Cache for
    the deeper layers;
  * three fp32 NHWC detection-conv outputs that the decode kernel turns into
    the (B, N, 5+C) prediction tensor in one launch.

Nothing here computes on the CPU; host code only builds descriptors.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import BnTrain, ConvDesc, HeadDesc, check, lib


class UnsupportedGraph(NotImplementedError):
    pass


def graph_from_blocks(blocks, channels, num_classes, num_anchors, detection_layers=()):
    """Per-block table: type, shapes-independent attributes, absolute sources.

    A convolution's width is its block's ``filters``.  The model constructor has
    already rewritten that entry for the caller's detection layers
    (/root/reference/yolo_v3.py:24-26); for blocks that come straight from a cfg,
    ``detection_layers`` names the blocks to widen to (num_classes + 5) * num_anchors."""
    det = (num_classes + 5) * num_anchors
    table, out_ch = [], []
    cin = channels
    n_yolo = 0
    for i, blk in enumerate(blocks):
        t = blk["type"]
        if t == "convolutional":
            cout = det if i in detection_layers else int(blk["filters"])
            k = int(blk["size"])
            ent = dict(type="conv", cin=cin, cout=cout, k=k, stride=int(blk["stride"]),
                       bn=bool("batch_normalize" in blk and blk["batch_normalize"]),
                       leaky=blk["activation"] == "leaky")
            cin = cout
        elif t == "upsample":
            ent = dict(type="upsample", stride=int(blk["stride"]))
        elif t == "route":
            items = [int(s) for s in blk["layers"].split(",")]
            srcs = [i + items[0]] + ([items[1]] if len(items) == 2 else [])
            if len(items) > 2:
                raise Exception("Route layer is not behaving as we planned, please "
                                "change the code according.")
            ent = dict(type="route", srcs=srcs)
            cin = sum(out_ch[s] for s in srcs)
        elif t == "shortcut":
            ent = dict(type="shortcut", src=i + int(blk["from"]))
        elif t == "yolo":
            ent = dict(type="yolo", head=n_yolo)
            n_yolo += 1
        else:
            raise UnsupportedGraph(f"block {i}: unknown type {t!r}")
        ent["cout"] = cin
        out_ch.append(cin)
        table.append(ent)
    return table


class _Val:
    """A tensor flowing between blocks: one or two NHWC parts."""
    __slots__ = ("parts", "H", "W")

    def __init__(self, parts, H, W):
        self.parts = parts        # list of (buffer_id, channels, up_log2)
        self.H, self.W = H, W


def build_ops(table, B, H, channels):
    """Pure host pass over the block table: which convolution reads what.

    Returns (ops, heads, buf_info):
      ops      one dict per convolution in execution order (sources, shapes,
               fused residual, fp32 output flag)
      heads    [(buffer id, G)] for the yolo blocks, in file order
      buf_info buffer id -> dict(bytes, first, last) with first/last the block
               indices between which the buffer must stay intact
    """
    vals = [None] * len(table)
    ops, heads, buf_info = [], [], {}
    cur = _Val([("input", int(channels), 0)], H, H)

    def touch(bid, i):
        if bid != "input":
            buf_info[bid]["last"] = max(buf_info[bid]["last"], i)

    i, n = 0, len(table)
    while i < n:
        ent = table[i]
        t = ent["type"]
        if t == "conv":
            k, s = ent["k"], ent["stride"]
            pad = k // 2
            Ho = (cur.H + 2 * pad - k) // s + 1
            Wo = (cur.W + 2 * pad - k) // s + 1
            if len(cur.parts) > 2:
                raise UnsupportedGraph(f"block {i}: more than two concatenated inputs")
            if len(cur.parts) == 2 and cur.parts[1][2] != 0:
                raise UnsupportedGraph(f"block {i}: only the first concat input may be upsampled")
            if cur.parts[0][2] not in (0, 1) or (cur.parts[0][2] == 1 and k != 1):
                raise UnsupportedGraph(f"block {i}: upsampled input into a {k}x{k} convolution")
            to_yolo = i + 1 < n and table[i + 1]["type"] == "yolo"
            fuse_res = i + 1 < n and table[i + 1]["type"] == "shortcut"
            op = dict(index=i, ent=ent, src=list(cur.parts), H=cur.H, W=cur.W, Ho=Ho, Wo=Wo,
                      fp32=to_yolo, res=None, first=(cur.parts[0][0] == "input"))
            for bid, _, _ in cur.parts:
                touch(bid, i)
            elem = 4 if to_yolo else 2
            bid = f"L{i}"
            buf_info[bid] = dict(bytes=B * Ho * Wo * ent["cout"] * elem, first=i, last=i)
            out = _Val([(bid, ent["cout"], 0)], Ho, Wo)
            vals[i] = out
            if fuse_res:
                sc = table[i + 1]
                rv = vals[sc["src"]]
                if rv is None or len(rv.parts) != 1 or rv.parts[0][2] != 0 or \
                        rv.parts[0][1] != ent["cout"] or rv.H != Ho or rv.W != Wo or to_yolo:
                    raise UnsupportedGraph(f"block {i + 1}: shortcut operand shape mismatch")
                op["res"] = rv.parts[0][0]
                touch(op["res"], i)
                vals[i + 1] = out          # the add happens in place in the conv output
                i += 1
            ops.append(op)
            cur = out
        elif t == "upsample":
            if ent["stride"] != 2 or len(cur.parts) != 1 or cur.parts[0][2] != 0:
                raise UnsupportedGraph(f"block {i}: only a single 2x nearest upsample is fused")
            bid, c, _ = cur.parts[0]
            cur = _Val([(bid, c, 1)], cur.H * 2, cur.W * 2)
            vals[i] = cur
        elif t == "route":
            srcs = [vals[s] for s in ent["srcs"]]
            if any(v is None for v in srcs):
                raise UnsupportedGraph(f"block {i}: route refers to a block with no tensor")
            if len(srcs) == 1:
                cur = srcs[0]
            else:
                a, b = srcs
                if a.H != b.H or a.W != b.W or len(a.parts) != 1 or len(b.parts) != 1:
                    raise UnsupportedGraph(f"block {i}: concat of mismatched tensors")
                cur = _Val([a.parts[0], b.parts[0]], a.H, a.W)
            vals[i] = cur
        elif t == "shortcut":
            raise UnsupportedGraph(f"block {i}: shortcut that does not follow a convolution")
        elif t == "yolo":
            if len(cur.parts) != 1 or not ops or ops[-1]["index"] != i - 1:
                raise UnsupportedGraph(f"block {i}: yolo block must follow its detection conv")
            bid = cur.parts[0][0]
            buf_info[bid]["last"] = 10 ** 9        # alive until the decode launch
            heads.append((bid, cur.H))
            vals[i] = cur
            if len(heads) == 3:
                break
        i += 1
    if not heads:
        raise UnsupportedGraph("cfg has no yolo block")
    return ops, heads, buf_info


def pack_conv_weights(wk, cout_pad):
    """[cout][K] weights (k = (kh*ks + kw)*cin + c, K a multiple of 32) -> the bf16
    K-tile-major image [K/32][cout_pad][32] the conv kernels stage from, zero rows
    for the padded output channels."""
    cout, K = wk.shape
    assert K % 32 == 0 and cout_pad >= cout
    wp = torch.zeros((cout_pad, K), dtype=torch.bfloat16, device=wk.device)
    wp[:cout] = wk.to(torch.bfloat16)
    return wp.view(cout_pad, K // 32, 32).permute(1, 0, 2).contiguous()


def place_buffers(buf_info):
    """First-fit placement by liveness, 256-byte aligned.  Returns
    ({buffer id: offset}, arena bytes)."""
    order = sorted(buf_info.items(), key=lambda kv: kv[1]["first"])
    placed = []            # (offset, end, last)
    offsets = {}
    peak = 0
    for bid, info in order:
        size = (info["bytes"] + 255) // 256 * 256
        live = sorted((o, e) for (o, e, last) in placed if last >= info["first"])
        pos = 0
        for o, e in live:
            if pos + size <= o:
                break
            pos = max(pos, e)
        offsets[bid] = pos
        placed.append((pos, pos + size, info["last"]))
        peak = max(peak, pos + size)
    return offsets, peak


def extend_fused_lifetimes(ops, buf_info, fuse_next):
    """fuse_next[j] > 0: ops j .. j + fuse_next[j] run as one launch.  Every output of
    the group is written while the group's first layer is still reading its inputs, so
    each is made alive from that layer on (place_buffers could otherwise put a later
    layer's output over the first layer's input, which dies with that layer)."""
    j = 0
    while j < len(ops):
        n = 1 + int(fuse_next[j])
        for o in ops[j + 1:j + n]:
            info = buf_info[f"L{o['index']}"]
            info["first"] = min(info["first"], ops[j]["index"])
        j += n


class Plan:
    """Descriptors and buffers for one (batch, height) shape on one device."""

    def __init__(self, model, B, H, keep_all=False, train=False):
        """keep_all=True gives every layer its own buffer (no reuse) so tests can
        inspect intermediate outputs after a run.

        train=True is the model in train() mode, which is how the callers that never
        call eval() run it (/root/reference/video.py:21-25, get_fps.py:25-29): every
        BatchNorm2d normalises with the statistics of the batch at hand and updates its
        running statistics, so nothing is folded; such a layer runs as convolution ->
        fp32 scratch -> statistics -> normalise + LeakyReLU + shortcut (csrc/bn.hip),
        and the module's running_mean / running_var / num_batches_tracked tensors are
        updated in place by the kernels."""
        _lib.require_gpu()
        self.keep_all = keep_all
        self.train = bool(train)
        self.model = model
        self.B, self.H = B, H
        params = model.params
        self.C = int(params["num_classes"])
        self.A = int(params["num_anchors"])
        self.D = 5 + self.C
        self.device = next(model.parameters()).device
        if self.device.type != "cuda":
            raise _lib.YoloAmdError("the model's parameters are on the CPU; call model.cuda() "
                                    "(params['cuda'] = True): the fused path runs on the GPU only")
        self.table = graph_from_blocks(model.blocks, int(params["channels"]), self.C, self.A)
        self.stage_events = None      # set to a list to collect per-stage HIP events
        self._build()

    # ------------------------------------------------------------------ build
    def _build(self):
        B, H = self.B, self.H
        dev = self.device
        ops, heads, buf_info = build_ops(self.table, B, H, int(self.model.params["channels"]))
        self.ops, self.heads = ops, heads
        # ---- descriptors: shapes first (the choice of fused launches below needs them, and
        # the placement of the buffers needs that choice), pointers once the buffers are placed
        self.descs = (ConvDesc * len(ops))()
        for j, op in enumerate(ops):
            ent, d = op["ent"], self.descs[j]
            (b0, c0, up0) = op["src"][0]
            d.c0, d.up0 = c0, up0
            d.c1 = op["src"][1][1] if len(op["src"]) == 2 else 0
            d.B, d.H, d.W = B, op["H"], op["W"]
            d.cout = ent["cout"]
            d.cout_pad = ent["cout"] if op["first"] else lib.yolo_amd_conv_cout_pad(ent["cout"])
            d.ksize, d.stride = ent["k"], ent["stride"]
            d.k_pad = ent["k"] * ent["k"] * (c0 + d.c1)
            d.leaky = int(ent["leaky"])
            d.dst_fp32 = int(op["fp32"])
            d.first_layer = int(op["first"])
        # The first layer and the stride-2 layer behind it run as one launch when
        # nothing else reads the first layer's output (its 64 bytes per input
        # pixel then never go to memory).  Plans that keep every layer readable
        # run them separately; YOLO_AMD_STEM=0 does too (development knob).
        self.fused_stem = False
        if (not self.keep_all and not self.train and len(ops) >= 2
                and os.environ.get("YOLO_AMD_STEM", "1") != "0"):
            o0, o1 = ops[0], ops[1]
            e0, e1 = o0["ent"], o1["ent"]
            readers = [op for op in ops if any(s_[0] == f"L{o0['index']}" for s_ in op["src"])
                       or op["res"] == f"L{o0['index']}"]
            if (o0["first"] and e0["k"] == 3 and e0["stride"] == 1 and e0["cout"] == 32 and e0["cin"] <= 3
                    and readers == [o1] and o1["src"] == [(f"L{o0['index']}", 32, 0)] and not o1["res"]
                    and e1["k"] == 3 and e1["stride"] == 2 and e1["cout"] == 64 and not o1["fp32"]):
                self.descs[0].fuse_next = 1
                self.fused_stem = True
                # ... and the 1x1 layer behind them (64 -> 32, the first residual block's
                # squeeze) is computed from the second layer's tile while it is in LDS
                # (fuse_next = 2: this and the next two descriptors); YOLO_AMD_STEM_TAIL=0
                # leaves it a launch of its own.
                if (len(ops) >= 3 and os.environ.get("YOLO_AMD_STEM_TAIL", "1") != "0"
                        and os.environ.get("YOLO_AMD_BLOCK", "0") != "1"):      # (that plan fuses the layer forwards)
                    o2 = ops[2]
                    e2 = o2["ent"]
                    if (e2["k"] == 1 and e2["stride"] == 1 and e2["cout"] == 32 and not o2["res"]
                            and o2["src"] == [(f"L{o1['index']}", 64, 0)] and not o2["fp32"]
                            and self.descs[2].cout_pad == 32):
                        self.descs[0].fuse_next = 2
        # The first residual block (1x1 64 -> 32, 3x3 32 -> 64, shortcut from the block's
        # input) CAN run as one launch when nothing else reads the 1x1 layer's output: its 32
        # channels then never go to memory and the block's input is read once (366 MB of HBM
        # traffic instead of 710 MB at batch 32 x 208^2, same bits).  Measured on MI355X the
        # one launch takes 165 us where the two take 56 + 95 us, and the benchmark loses 1.5 %
        # with it (the fused kernel's tile is a chain of four barrier-separated phases at two
        # waves per SIMD, and neither of the two layers it replaces was far from the HBM rate),
        # so it is off unless YOLO_AMD_BLOCK=1; it stays in the library and under test.
        self.fused_block = False
        if not self.keep_all and not self.train and os.environ.get("YOLO_AMD_BLOCK", "0") == "1":
            for j in range(len(ops) - 1):
                o0, o1 = ops[j], ops[j + 1]
                e0, e1 = o0["ent"], o1["ent"]
                mid = f"L{o0['index']}"
                readers = [op for op in ops if any(s_[0] == mid for s_ in op["src"]) or op["res"] == mid]
                if (not o0["first"] and e0["k"] == 1 and e0["stride"] == 1 and e0["cin"] == 64 and e0["cout"] == 32
                        and len(o0["src"]) == 1 and o0["src"][0][2] == 0 and not o0["res"] and not o0["fp32"]
                        and readers == [o1] and o1["src"] == [(mid, 32, 0)]
                        and e1["k"] == 3 and e1["stride"] == 1 and e1["cout"] == 64 and not o1["fp32"]
                        and o1["res"] == o0["src"][0][0] and self.descs[j].cout_pad == 32
                        and self.descs[j + 1].cout_pad == 64):
                    self.descs[j].fuse_next = 1
                    self.fused_block = True
                    break
        # A 3x3 layer with 128 output channels whose kernel holds all of them for a pixel tile
        # in one workgroup (the im2col kernel's 128-channel tile with the two-step activation
        # tile: the stride-2 layer in front of the second group of residual blocks; the flat
        # patch-reuse kernel: the 3x3 layer of the first block of that group) computes the
        # 1x1 layer behind it (128 -> 64) from its output tile in LDS: fuse_next on the 3x3 layer,
        # both outputs written, same bits; the library runs the pair as two launches when the
        # kernel it picks for the 3x3 layer cannot.  YOLO_AMD_TAIL=0: never (development knob).
        self.fused_tails = 0
        if not self.keep_all and not self.train and os.environ.get("YOLO_AMD_TAIL", "1") != "0":
            for j in range(1, len(ops) - 1):
                o0, o1 = ops[j], ops[j + 1]
                e0, e1 = o0["ent"], o1["ent"]
                if (self.descs[j].fuse_next or self.descs[j - 1].fuse_next or self.descs[j + 1].fuse_next
                        or (j >= 2 and self.descs[j - 2].fuse_next == 2)):
                    continue
                if (e0["k"] == 3 and e0["cout"] == 128 and not o0["first"] and not o0["fp32"]
                        and e1["k"] == 1 and e1["stride"] == 1 and e1["cout"] == 64 and not o1["res"] and not o1["fp32"]
                        and o1["src"] == [(f"L{o0['index']}", 128, 0)]
                        and self.descs[j].cout_pad == 128 and self.descs[j + 1].cout_pad == 64):
                    ch = _lib.conv_choice(self.descs[j])
                    if ((ch["kernel"] == "igemm" and ch["act_ksteps"] == 2 and ch["bn"] == 128 and ch["k_slices"] == 1)
                            or (ch["kernel"] == "patch" and ch["patch_flat"] and ch["bn"] == 128)):
                        self.descs[j].fuse_next = 1
                        self.fused_tails += 1
        # ---- buffers.  A launch that runs several layers writes the later layers' outputs
        # while the first layer's inputs are still being read: those outputs are alive from
        # the first layer of the group on, or they could be placed over that input.
        extend_fused_lifetimes(ops, buf_info, [int(d.fuse_next) for d in self.descs])
        if self.keep_all:
            for info in buf_info.values():
                info["last"] = 10 ** 9
        offsets, peak = place_buffers(buf_info)
        self.arena = torch.empty(peak + 256, dtype=torch.uint8, device=dev)
        self.arena_bytes = peak
        base = (self.arena.data_ptr() + 255) // 256 * 256
        self.buf_ptr = {bid: base + off for bid, off in offsets.items()}
        self.zero_page = torch.zeros(1024, dtype=torch.uint8, device=dev)
        for j, op in enumerate(ops):
            d = self.descs[j]
            b0 = op["src"][0][0]
            d.src0 = 0 if b0 == "input" else self.buf_ptr[b0]
            d.src1 = self.buf_ptr[op["src"][1][0]] if len(op["src"]) == 2 else None
            d.residual = self.buf_ptr[op["res"]] if op["res"] else None
            d.dst = self.buf_ptr[f"L{op['index']}"]
        # Scratch for split reductions: at small batch the deep layers have a few
        # output tiles and K up to 4608; the library cuts such a K into slices when
        # it is lent a buffer (one for all layers: they run in stream order).
        need = max((lib.yolo_amd_conv_splitk_bytes(ctypes.byref(self.descs[j])) for j in range(len(ops))),
                   default=0)
        self.splitk_bytes = int(need)
        if need:
            self._splitk = torch.empty(need, dtype=torch.uint8, device=dev)
            for j in range(len(ops)):
                self.descs[j].splitk_ws = self._splitk.data_ptr()
                self.descs[j].splitk_ws_bytes = need
        # Batch-statistics BatchNorm: one fp32 scratch for the largest un-normalised
        # convolution output, one for the partial sums, one for (scale, shift); the
        # layers run in stream order and share them.
        self._bn = None
        if self.train:
            bn_ops = [op for op in ops if op["ent"]["bn"]]
            for op in bn_ops:
                if op["fp32"] or op["ent"]["cout"] % 8:
                    raise UnsupportedGraph(f"block {op['index']}: batch-statistics BatchNorm needs a bf16 output "
                                           "with a multiple of 8 channels")
                if B * op["Ho"] * op["Wo"] <= 1:
                    raise ValueError("Expected more than 1 value per channel when training")
            if bn_ops:
                raw_elems = max(B * op["Ho"] * op["Wo"] * op["ent"]["cout"] for op in bn_ops)
                cmax = max(op["ent"]["cout"] for op in bn_ops)
                self._bn_raw = torch.empty(raw_elems, dtype=torch.float32, device=dev)
                self._bn_partial = torch.empty(_lib.BN_MAX_PARTIALS * cmax * 2, dtype=torch.float64, device=dev)
                self._bn_scale_shift = torch.empty(2 * cmax, dtype=torch.float32, device=dev)
                self._bn = (BnTrain * len(ops))()
        self.refold()

        # ---- decode descriptors
        stride_rows = 0
        self.head_descs = (HeadDesc * len(heads))()
        anchors = np.asarray(self.model.params["anchors"], dtype=np.float64)
        for k, (bid, G) in enumerate(heads):
            hd = self.head_descs[k]
            hd.raw = self.buf_ptr[bid]
            hd.G = G
            hd.layout = _lib.LAYOUT_NHWC
            hd.row_offset = stride_rows
            stride = H // G
            hd.stride = float(stride)
            # anchors may have been reassigned on the yolo_layer modules
            layer_anchors = self._head_anchors(k, anchors)
            for a in range(self.A):
                hd.anchor_w[a] = np.float32(layer_anchors[a][0] / stride)
                hd.anchor_h[a] = np.float32(layer_anchors[a][1] / stride)
            stride_rows += G * G * self.A
        self.N = stride_rows
        # The detection convolutions decode in their epilogue and write straight
        # into the prediction tensor; the separate decode launch remains for plans
        # that keep the raw head outputs readable (keep_all).
        self.fused_decode = not self.keep_all
        self._head_slots = []
        if self.fused_decode:
            for k, (bid, G) in enumerate(heads):
                j = next(j for j, op in enumerate(ops) if f"L{op['index']}" == bid)
                d = self.descs[j]
                d.decode = ctypes.pointer(self.head_descs[k])
                d.decode_rows = self.N
                d.decode_anchors = self.A
                self._head_slots.append(j)

    def _head_anchors(self, k, anchors):
        idx = [i for i, e in enumerate(self.table) if e["type"] == "yolo"][k]
        mod = self.model.module_list[idx]
        layer = mod[0] if len(mod) else None
        if layer is not None and getattr(layer, "anchors", None) is not None:
            return np.asarray(layer.anchors, dtype=np.float64)
        return anchors[k * self.A:(k + 1) * self.A]

    # ------------------------------------------------------------------ weights
    @torch.no_grad()
    def refold(self):
        """(Re)compute folded weights from the model's current parameters."""
        self._packed = []
        for j, op in enumerate(self.ops):
            i, ent, d = op["index"], op["ent"], self.descs[j]
            seq = self.model.module_list[i]
            conv = seq[0]
            w = conv.weight.detach().float()
            cout = w.shape[0]
            if cout != ent["cout"] or w.shape[1] != ent["cin"] or w.shape[2] != ent["k"]:
                raise _lib.YoloAmdError(f"block {i}: the module holds a {tuple(w.shape)} convolution, the plan "
                                        f"was laid out for {ent['cout']}x{ent['cin']}x{ent['k']}x{ent['k']}")
            if ent["bn"] and self.train:
                bias = conv.bias.detach().float() if conv.bias is not None else w.new_zeros(cout)
                self._fill_bn(j, i, seq[1])
            elif ent["bn"]:
                bn = seq[1]
                scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.float() + bn.eps)
                bias = bn.bias.detach().float() - bn.running_mean.float() * scale
                if conv.bias is not None:
                    bias = bias + conv.bias.detach().float() * scale
                w = w * scale.view(-1, 1, 1, 1)
            else:
                bias = conv.bias.detach().float() if conv.bias is not None else w.new_zeros(cout)
            wk = w.permute(0, 2, 3, 1).reshape(cout, -1)          # [cout][(kh,kw,cin)]
            if op["first"]:
                wp = wk.t().contiguous()                           # fp32 [k][cout]
                bp = bias.contiguous()
            else:
                wp = pack_conv_weights(wk, d.cout_pad)
                bp = torch.zeros(d.cout_pad, dtype=torch.float32, device=w.device)
                bp[:cout] = bias
            self._packed.append((wp, bp))
            d.weight = wp.data_ptr()
            d.bias = bp.data_ptr()

    def _fill_bn(self, j, index, bn):
        """Point layer j's batch-statistics descriptor at the module's own tensors."""
        def ptr(t, dtype=torch.float32):
            if t is None:
                return None
            if t.dtype != dtype or t.device != self.device or not t.is_contiguous():
                raise UnsupportedGraph(f"block {index}: BatchNorm tensors must be contiguous {dtype} on {self.device}")
            return t.data_ptr()
        if bn.weight is None or bn.bias is None:
            raise UnsupportedGraph(f"block {index}: BatchNorm2d without affine parameters")
        b = self._bn[j]
        b.raw = self._bn_raw.data_ptr()
        b.partial = self._bn_partial.data_ptr()
        b.scale_shift = self._bn_scale_shift.data_ptr()
        b.gamma, b.beta = ptr(bn.weight.data), ptr(bn.bias.data)
        track = bn.track_running_stats and bn.running_mean is not None
        b.running_mean = ptr(bn.running_mean) if track else None
        b.running_var = ptr(bn.running_var) if track else None
        b.num_batches_tracked = ptr(bn.num_batches_tracked, torch.int64) if track else None
        b.eps = float(bn.eps)
        b.momentum = -1.0 if bn.momentum is None else float(bn.momentum)
        self.descs[j].bn_train = ctypes.pointer(b)

    # ------------------------------------------------------------------ run
    def run(self, x):
        """x: (B, C, H, H) fp32 contiguous on the plan's device.  Returns the
        (B, N, 5+C) fp32 prediction tensor (asynchronous on the current stream)."""
        if x.shape[0] != self.B or x.shape[2] != self.H or x.shape[3] != self.H:
            raise ValueError("input shape does not match the plan")
        x = x.contiguous()
        if x.dtype != torch.float32:
            x = x.float()
        self.descs[0].src0 = x.data_ptr()
        out = torch.empty((self.B, self.N, self.D), dtype=torch.float32, device=self.device)
        for j in self._head_slots:
            self.descs[j].dst = out.data_ptr()
        with torch.cuda.device(self.device):
            stream = _lib.current_stream_ptr(self.device)
            ev = None
            if self.stage_events is not None:
                # HIP events on the launch stream around the conv stack and the
                # decode launch (read back by bench.py after the timed region)
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                ev[0].record()
            check(lib.yolo_amd_conv_forward(self.descs, len(self.ops), self.zero_page.data_ptr(),
                                            stream), "conv_forward")
            if ev:
                ev[1].record()
            if not self.fused_decode:
                check(lib.yolo_amd_decode(self.head_descs, len(self.heads), self.B, self.A, self.D,
                                          self.N, out.data_ptr(), stream), "decode")
            if ev:
                ev[2].record()
                self.stage_events.append(ev)
        self._keepalive = x
        return out

    def layer_output(self, index):
        """Debug/test view of a conv output buffer as an NHWC tensor."""
        for op in self.ops:
            if op["index"] == index:
                dt = torch.float32 if op["fp32"] else torch.bfloat16
                n = self.B * op["Ho"] * op["Wo"] * op["ent"]["cout"]
                esize = 4 if op["fp32"] else 2
                off = self.buf_ptr[f"L{index}"] - self.arena.data_ptr()
                flat = self.arena[off:off + n * esize].view(dt)
                return flat.view(self.B, op["Ho"], op["Wo"], op["ent"]["cout"])
        raise KeyError(index)
