"""Helpers for the GPU tests: build the drop-in model with synthetic weights,
run single fused conv layers through the C ABI."""
from __future__ import annotations

import ctypes
import os
import tempfile

import numpy as np
import torch

import gen_inputs as gi
import yolo_oracle as orc
from yolo_amd.cfg import default_params, write_yolov3_cfg, yolov3_cfg_text


def set_knob(monkeypatch, name, value):
    """Set a YOLO_AMD_* development knob for the rest of the test: the library caches
    the environment, so it is told to read it again (conftest does the same after
    every test, once monkeypatch has restored the variable)."""
    from yolo_amd._lib import lib
    monkeypatch.setenv(name, value)
    lib.yolo_amd_reload_knobs()


def require_gpu():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device (run with -m 'not gpu' on CPU)"


def synthetic_snapshot(C, A, seed):
    """Write the seeded synthetic state_dict to a .pth and return (path, sd)."""
    blocks = orc.parse_cfg_text(yolov3_cfg_text(C, A))
    table = orc.layer_table(blocks, 3, C, A)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in gi.synthetic_state_dict(table, seed).items()}
    fd, path = tempfile.mkstemp(prefix="yolo_amd_", suffix=".pth")
    os.close(fd)
    torch.save(sd, path)
    return path, sd


def build_model(H, C=80, A=3, seed=3, confidence=0.5, nms=0.4):
    """The drop-in yolo_v3 on the GPU, eval mode, synthetic weights loaded
    through params['pretrain_snapshot'] like a real checkpoint."""
    from yolo_amd.utilis import parse_cfg
    from yolo_amd.yolo_v3 import yolo_v3
    require_gpu()
    cfg = write_yolov3_cfg(None, C, A)
    snap, sd = synthetic_snapshot(C, A, seed)
    params = default_params(H, C, A, cuda=True, confidence=confidence, nms_thesh=nms)
    params["pretrain_snapshot"] = snap
    blocks = parse_cfg(cfg)
    model = yolo_v3(params, blocks)
    model.eval()
    os.remove(cfg)
    os.remove(snap)
    return model, params, sd


def oracle_forward(sd, H, C, A, x_np, return_layers=False, emulate_bf16=False):
    """The CPU oracle's forward pass: fp32 as the reference computes it, or
    (emulate_bf16) with the storage roundings of the bf16 device path put in."""
    blocks = orc.parse_cfg_text(yolov3_cfg_text(C, A))
    p = default_params(H, C, A)
    return orc.forward(sd, blocks, p, torch.from_numpy(x_np), return_layers=return_layers,
                       emulate_bf16=emulate_bf16)


def run_conv(x0, w, bias, *, stride=1, leaky=True, x1=None, up0=0, residual=None, fp32_out=False,
             splitk=False, info=None, bn=None, repeat=1):
    """One fused layer through yolo_amd_conv_forward.  splitk: lend the scratch the
    library asks for, so a layer with few output tiles splits its reduction; info (a
    dict) receives "splitk_bytes" and "choice" (the kernel the library picks, as
    yolo_amd_conv_query reports it).

    x0: (B, H0, W0, c0) bf16 NHWC cuda; x1: optional (B, H, W, c1) bf16;
    w: (cout, cin, k, k) fp32 (already BN-folded); bias: (cout,) fp32.
    bn: dict(gamma, beta, running_mean, running_var, num_batches_tracked, eps, momentum) of
    CUDA tensors / floats: BatchNorm2d on batch statistics behind the convolution (w then
    holds the convolution's own weights); the running tensors are updated in place.
    repeat > 1: the same descriptor (and scratch) is launched again that many times; every
    launch must write the same bits.
    Returns the NHWC output tensor (bf16, or fp32 when fp32_out)."""
    from yolo_amd import _lib
    from yolo_amd._lib import ConvDesc, lib, check
    dev = x0.device
    cout, cin, k, _ = w.shape
    B = x0.shape[0]
    H, W = x0.shape[1] << up0, x0.shape[2] << up0
    c0 = x0.shape[3]
    c1 = 0 if x1 is None else x1.shape[3]
    assert c0 + c1 == cin
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    cout_pad = lib.yolo_amd_conv_cout_pad(cout)
    K = k * k * cin
    from yolo_amd.engine import pack_conv_weights
    wp = pack_conv_weights(w.permute(0, 2, 3, 1).reshape(cout, K).to(dev), cout_pad)
    bp = torch.zeros(cout_pad, dtype=torch.float32, device=dev)
    bp[:cout] = bias.to(dev)
    out = torch.full((B, Ho, Wo, cout), float("nan"), dtype=torch.float32 if fp32_out else torch.bfloat16,
                     device=dev)
    zero = torch.zeros(1024, dtype=torch.uint8, device=dev)
    d = (ConvDesc * 1)()
    d[0].src0 = x0.data_ptr()
    d[0].src1 = None if x1 is None else x1.data_ptr()
    d[0].weight, d[0].bias = wp.data_ptr(), bp.data_ptr()
    d[0].residual = None if residual is None else residual.data_ptr()
    d[0].dst = out.data_ptr()
    d[0].c0, d[0].c1, d[0].up0 = c0, c1, up0
    d[0].B, d[0].H, d[0].W = B, H, W
    d[0].cout, d[0].cout_pad, d[0].ksize, d[0].stride = cout, cout_pad, k, stride
    d[0].k_pad, d[0].leaky, d[0].dst_fp32, d[0].first_layer = K, int(leaky), int(fp32_out), 0
    need = int(lib.yolo_amd_conv_splitk_bytes(d))
    if info is not None:
        info["splitk_bytes"] = need
    if splitk and need:
        ws = torch.full((need,), 0xFF, dtype=torch.uint8, device=dev)      # NaN-filled scratch
        d[0].splitk_ws, d[0].splitk_ws_bytes = ws.data_ptr(), need
    if bn is not None:
        b = _lib.BnTrain()
        keep = [torch.full((B * Ho * Wo * cout,), float("nan"), dtype=torch.float32, device=dev),
                torch.full((_lib.BN_MAX_PARTIALS * cout * 2,), float("nan"), dtype=torch.float64, device=dev),
                torch.full((2 * cout,), float("nan"), dtype=torch.float32, device=dev)]
        b.raw, b.partial, b.scale_shift = (t.data_ptr() for t in keep)
        b.gamma, b.beta = bn["gamma"].data_ptr(), bn["beta"].data_ptr()
        for key in ("running_mean", "running_var", "num_batches_tracked"):
            setattr(b, key, bn[key].data_ptr() if bn.get(key) is not None else None)
        b.eps = bn.get("eps", 1e-5)
        b.momentum = -1.0 if bn.get("momentum", 0.1) is None else bn.get("momentum", 0.1)
        d[0].bn_train = ctypes.pointer(b)
    if info is not None:
        info["choice"] = _lib.conv_choice(d[0])      # which kernel the library runs for this layer
    check(lib.yolo_amd_conv_forward(d, 1, zero.data_ptr(), _lib.current_stream_ptr(dev)), "conv")
    torch.cuda.synchronize()
    for _ in range(repeat - 1):
        assert bn is None, "a train-mode layer moves its running statistics on every launch"
        first = out.clone()
        out.fill_(float("nan"))
        check(lib.yolo_amd_conv_forward(d, 1, zero.data_ptr(), _lib.current_stream_ptr(dev)), "conv")
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.uint8), first.view(torch.uint8)), "a second launch wrote different bits"
    return out


def conv_reference(x0, w, bias, *, stride=1, leaky=True, x1=None, up0=0, residual=None):
    """fp32 torch-CPU reference on the SAME bf16-rounded operands (NHWC in/out)."""
    import torch.nn.functional as F
    a = x0.float().cpu().permute(0, 3, 1, 2)
    if up0:
        a = F.interpolate(a, scale_factor=2, mode="nearest")
    if x1 is not None:
        a = torch.cat((a, x1.float().cpu().permute(0, 3, 1, 2)), 1)
    wq = w.to(torch.bfloat16).float().cpu()
    y = F.conv2d(a, wq, bias.float().cpu(), stride=stride, padding=w.shape[2] // 2)
    if leaky:
        y = F.leaky_relu(y, 0.1)
    if residual is not None:
        y = y + residual.float().cpu().permute(0, 3, 1, 2)
    return y.permute(0, 2, 3, 1).contiguous()


def case_seed(name):
    """A fixed seed per test-case name (str hashes are salted per process)."""
    import zlib
    return zlib.crc32(name.encode()) % 10000


def keep_set(pred, rows):
    """The (image, row) pairs of `pred` (B, N, D) that the (K, 9) filter_results rows
    stand for: a row's corner box is a pure function of one prediction row, so the
    pairs are recovered by matching the four corner floats bit for bit."""
    out = set()
    if rows.shape[0] == 0:
        return out
    p = pred.astype(np.float32)
    half_w, half_h = p[..., 2] / np.float32(2), p[..., 3] / np.float32(2)
    corners = np.stack([p[..., 0] - half_w, p[..., 1] - half_h, p[..., 0] + half_w, p[..., 1] + half_h], -1)
    for b in np.unique(rows[:, 0]).astype(int):
        index = {}
        for n, c in enumerate(corners[b].view(np.uint32)):
            index.setdefault(c.tobytes(), n)
        for r in rows[rows[:, 0] == b][:, 1:5].astype(np.float32).view(np.uint32):
            n = index.get(r.tobytes())
            assert n is not None, "an output row matches no prediction row"
            out.add((b, n))
    return out


def record_metric(name, values):
    """Append one entry to gpurun_out/parity_metrics.json (copied into profiles/ by
    hand after a GPU run): what the bf16 path measured against the fp32 oracle."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "gpurun_out", "parity_metrics.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = {}
    if os.path.exists(path):
        try:
            with open(path) as fh:
                data = json.load(fh)
        except Exception:
            data = {}
    data[name] = values
    with open(path, "w") as fh:
        json.dump(data, fh, indent=1, sort_keys=True)


def teacher_forced_layers(keep, outs16):
    """Every fused conv launch of a keep_all plan on the ORACLE's activations.

    outs16: the per-block outputs of the oracle with bf16 storage (emulate_bf16=True) for
    the same input the plan has just run on.  For each launch the buffers it reads (sources,
    residual operand) are overwritten with the oracle's tensors -- bf16 values, so nothing is
    lost -- the launch runs alone, and its output is compared with the oracle's output of the
    same fused layer.  Same bf16 operands in, so all that may differ is the order of the fp32
    sum: an element either has the oracle's bits or sits on the neighbouring bf16 value
    because the two fp32 sums fell on different sides of a rounding boundary.

    Returns one dict per launch: block, mismatch (share of elements whose bits differ),
    rel_l2, worst (largest |got - ref| / (2^-7 |ref| + 1e-4 rms(ref)): <= 1 means every
    differing element is within one bf16 step), and for fp32 outputs max_rel instead."""
    import ctypes
    from yolo_amd import _lib
    from yolo_amd._lib import check, lib
    by_index = {op["index"]: op for op in keep.ops}

    def ref_of(index):          # what the buffer of block `index` holds: after the fused shortcut add
        return outs16[index + 1] if by_index[index]["res"] else outs16[index]

    def put(index):
        ref = ref_of(index).permute(0, 2, 3, 1).contiguous()
        buf = keep.layer_output(index)
        buf.copy_(ref.to(buf.dtype).cuda())

    size = ctypes.sizeof(_lib.ConvDesc)
    base = ctypes.addressof(keep.descs)
    stream = _lib.current_stream_ptr(keep.device)
    stats = []
    for j, op in enumerate(keep.ops):
        for bid, _, _ in op["src"]:
            if bid != "input":
                put(int(bid[1:]))
        if op["res"]:
            put(int(op["res"][1:]))
        keep.layer_output(op["index"]).fill_(float("nan"))
        ptr = ctypes.cast(base + j * size, ctypes.POINTER(_lib.ConvDesc))
        check(lib.yolo_amd_conv_forward(ptr, 1, keep.zero_page.data_ptr(), stream), "conv_forward")
        torch.cuda.synchronize()
        got = keep.layer_output(op["index"])
        ref = ref_of(op["index"]).permute(0, 2, 3, 1).contiguous().cuda()
        g32 = got.float()
        rms = float(ref.double().pow(2).mean().sqrt())
        diff = (g32 - ref).abs()
        ent = dict(block=op["index"], k=op["ent"]["k"], cout=op["ent"]["cout"],
                   rel_l2=float(torch.linalg.vector_norm(diff.flatten(), dtype=torch.float64) /
                                (torch.linalg.vector_norm(ref.flatten(), dtype=torch.float64) + 1e-30)))
        assert bool(torch.isfinite(g32).all()), f"block {op['index']}: the launch left NaN / inf"
        if op["fp32"]:
            ent["max_rel"] = float((diff / (ref.abs() + 1e-3 * rms)).max())
        else:
            ent["mismatch"] = float((got.view(torch.int16) != ref.to(torch.bfloat16).view(torch.int16)).float().mean())
            ent["worst"] = float((diff / (ref.abs() * 2.0 ** -7 + 1e-4 * rms)).max())
        stats.append(ent)
        del got, ref, g32, diff
    return stats
