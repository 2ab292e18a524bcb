"""The model in train() mode on the HIP path: BatchNorm2d on batch statistics behind the
MFMA convolutions (csrc/bn.hip), as the callers that never call eval() run the reference
(/root/reference/video.py:21-25, get_fps.py:25-29).

Single layers go through the C ABI against torch's fp32 batch_norm(training=True) on the
same bf16-rounded operands; whole forward passes against the vectors the reference itself
returned in train() mode (tests/golden/make_golden_train.py) and against the oracle."""
import json
import os
import warnings

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import gen_inputs as gi
import yolo_oracle as orc
from gpu_helpers import build_model, case_seed, run_conv
from yolo_amd.cfg import default_params, yolov3_cfg_text

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRAIN_META = json.load(open(os.path.join(GOLD, "golden_train.json")))
TRAIN_ARR = np.load(os.path.join(GOLD, "golden_train.npz"))

# name, B, H, cin, cout, k, stride, residual, upsample+concat, momentum, expected kernel
LAYER_CASES = [
    ("1x1 64->32", 2, 20, 64, 32, 1, 1, False, False, 0.1, "igemm"),
    ("3x3 32->64 residual", 2, 24, 32, 64, 3, 1, True, False, 0.1, "igemm"),      # fp32 scratch: not the thin kernel
    ("3x3 stride 2 64->128", 3, 30, 64, 128, 3, 2, False, False, 0.1, "igemm"),
    ("3x3 128->256 patch kernel", 16, 26, 128, 256, 3, 1, True, False, 0.1, "patch"),
    ("1x1 upsample+concat 384->128", 2, 16, 384, 128, 1, 1, False, True, 0.1, "igemm"),
    ("3x3 512->1024 one image", 1, 13, 512, 1024, 3, 1, True, False, 0.1, "igemm"),
    ("1x1 cumulative average", 2, 12, 128, 64, 1, 1, False, False, None, "igemm"),
    ("1x1 ragged 5x7x9 pixels", 5, 7, 96, 40, 1, 1, False, False, 0.3, "igemm"),
]


def _bn_tensors(r, cout, momentum, nbt0=0):
    return dict(gamma=torch.from_numpy(r.uniform(0.5, 1.5, cout).astype(np.float32)).cuda(),
                beta=torch.from_numpy(r.normal(0, 0.3, cout).astype(np.float32)).cuda(),
                running_mean=torch.from_numpy(r.normal(0, 0.2, cout).astype(np.float32)).cuda(),
                running_var=torch.from_numpy(r.uniform(0.5, 1.5, cout).astype(np.float32)).cuda(),
                num_batches_tracked=torch.tensor(nbt0, dtype=torch.int64, device="cuda"),
                eps=1e-5, momentum=momentum)


def _bn_reference(y, bn, leaky, residual):
    """y: NCHW fp32 conv output (CPU).  torch's own training-mode batch_norm."""
    rm, rv = bn["running_mean"].cpu().clone(), bn["running_var"].cpu().clone()
    nbt = int(bn["num_batches_tracked"]) + 1
    mom = bn["momentum"] if bn["momentum"] is not None else 1.0 / nbt
    out = F.batch_norm(y, rm, rv, bn["gamma"].cpu(), bn["beta"].cpu(), training=True, momentum=mom, eps=bn["eps"])
    if leaky:
        out = F.leaky_relu(out, 0.1)
    if residual is not None:
        out = out + residual.float().cpu().permute(0, 3, 1, 2)
    return out.permute(0, 2, 3, 1).contiguous(), rm, rv, nbt


@pytest.mark.parametrize("case", LAYER_CASES, ids=lambda c: c[0])
def test_conv_with_batch_statistics(case):
    name, B, H, cin, cout, k, stride, use_res, upcat, momentum, kernel = case
    r = gi.rs(case_seed("bn " + name))
    w = torch.from_numpy((r.standard_normal((cout, cin, k, k)) * np.sqrt(2.0 / (cin * k * k))).astype(np.float32))
    cbias = torch.from_numpy(r.normal(0, 0.2, cout).astype(np.float32))     # a convolution bias in front of the BN
    if upcat:
        c0 = cin // 3
        x0 = torch.from_numpy(r.standard_normal((B, H // 2, H // 2, c0)).astype(np.float32)).cuda().to(torch.bfloat16)
        x1 = torch.from_numpy(r.standard_normal((B, H, H, cin - c0)).astype(np.float32)).cuda().to(torch.bfloat16)
        a = torch.cat((F.interpolate(x0.float().cpu().permute(0, 3, 1, 2), scale_factor=2, mode="nearest"),
                       x1.float().cpu().permute(0, 3, 1, 2)), 1)
    else:
        x0 = torch.from_numpy((r.standard_normal((B, H, H, cin)) + 0.3).astype(np.float32)).cuda().to(torch.bfloat16)
        x1 = None
        a = x0.float().cpu().permute(0, 3, 1, 2)
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    res = None
    if use_res:
        res = torch.from_numpy(r.standard_normal((B, Ho, Ho, cout)).astype(np.float32)).cuda().to(torch.bfloat16)
    bn = _bn_tensors(r, cout, momentum, nbt0=3)
    y = F.conv2d(a, w.to(torch.bfloat16).float(), cbias, stride=stride, padding=k // 2)
    want, rm, rv, nbt = _bn_reference(y, bn, True, res)
    info = {}
    got = run_conv(x0, w, cbias, stride=stride, leaky=True, x1=x1, up0=int(upcat), residual=res, bn=bn,
                   splitk=True, info=info)
    assert info["choice"]["kernel"] == kernel, info["choice"]
    assert got.dtype == torch.bfloat16
    torch.testing.assert_close(got.float().cpu(), want, rtol=1.0 / 128, atol=1.0 / 128)
    torch.testing.assert_close(bn["running_mean"].cpu(), rm, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(bn["running_var"].cpu(), rv, rtol=2e-5, atol=2e-6)
    assert int(bn["num_batches_tracked"]) == nbt
    # a second call moves the statistics again and gives the same output bits
    got2 = run_conv(x0, w, cbias, stride=stride, leaky=True, x1=x1, up0=int(upcat), residual=res, bn=bn, splitk=True)
    assert torch.equal(got2.view(torch.int16), got.view(torch.int16)), "the statistics are not deterministic"
    assert int(bn["num_batches_tracked"]) == nbt + 1


@pytest.mark.parametrize("cout", [32, 16])
def test_first_layer_with_batch_statistics(cout):
    """The network input layer (NCHW fp32 in, hi/lo bf16 split): fp32 scratch output."""
    from yolo_amd import _lib
    from yolo_amd._lib import ConvDesc, check, lib
    import ctypes
    r = gi.rs(50 + cout)
    B, H = 2, 37
    x = torch.from_numpy(gi.images(5, B, H)).cuda()
    w = torch.from_numpy((r.standard_normal((cout, 3, 3, 3)) * 0.27).astype(np.float32))
    zero_bias = torch.zeros(cout, dtype=torch.float32, device="cuda")
    wk = w.permute(0, 2, 3, 1).reshape(cout, 27).t().contiguous().cuda()
    bn = _bn_tensors(r, cout, 0.1)
    out = torch.full((B, H, H, cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    keep = [torch.full((B * H * H * cout,), float("nan"), dtype=torch.float32, device="cuda"),
            torch.empty(_lib.BN_MAX_PARTIALS * cout * 2, dtype=torch.float64, device="cuda"),
            torch.empty(2 * cout, dtype=torch.float32, device="cuda")]
    b = _lib.BnTrain()
    b.raw, b.partial, b.scale_shift = (t.data_ptr() for t in keep)
    b.gamma, b.beta = bn["gamma"].data_ptr(), bn["beta"].data_ptr()
    b.running_mean, b.running_var = bn["running_mean"].data_ptr(), bn["running_var"].data_ptr()
    b.num_batches_tracked = bn["num_batches_tracked"].data_ptr()
    b.eps, b.momentum = 1e-5, 0.1
    d = (ConvDesc * 1)()
    d[0].src0, d[0].weight, d[0].bias, d[0].dst = x.data_ptr(), wk.data_ptr(), zero_bias.data_ptr(), out.data_ptr()
    d[0].c0, d[0].c1, d[0].up0, d[0].B, d[0].H, d[0].W = 3, 0, 0, B, H, H
    d[0].cout, d[0].cout_pad, d[0].ksize, d[0].stride, d[0].k_pad = cout, cout, 3, 1, 27
    d[0].leaky, d[0].dst_fp32, d[0].first_layer = 1, 0, 1
    d[0].bn_train = ctypes.pointer(b)
    want, rm, rv, nbt = _bn_reference(F.conv2d(x.cpu(), w, None, padding=1), bn, True, None)
    check(lib.yolo_amd_conv_forward(d, 1, None, _lib.current_stream_ptr(x.device)), "first conv + bn")
    torch.cuda.synchronize()
    torch.testing.assert_close(out.float().cpu(), want, rtol=1.0 / 128, atol=1.0 / 128)
    # the un-normalised sums keep fp32-level accuracy (hi/lo split), and so do the statistics
    raw = keep[0].view(B, H, H, cout).cpu()
    torch.testing.assert_close(raw, F.conv2d(x.cpu(), w, None, padding=1).permute(0, 2, 3, 1), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(bn["running_mean"].cpu(), rm, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(bn["running_var"].cpu(), rv, rtol=1e-4, atol=1e-5)
    assert int(bn["num_batches_tracked"]) == 1


def _oracle_state(C, A, seed):
    blocks = orc.parse_cfg_text(yolov3_cfg_text(C, A))
    table = orc.layer_table(blocks, 3, C, A)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in gi.synthetic_state_dict(table, seed).items()}
    return blocks, sd


# bf16 activations against the fp32 reference.  In train() mode every layer is re-normalised
# to unit variance on the statistics of as few as 8 values per channel (2x2 maps at 64
# pixels), so the damping of the residual branches that the running statistics carry in
# eval() mode is gone and bf16 rounding accumulates further: measured median 0.7-1.2e-2,
# p99 6-8e-2, objectness within 0.06 (eval() mode: 4-5e-3, 3-4e-2, 0.03).
GATE_MED, GATE_P99, GATE_OBJ = 2e-2, 0.12, 0.09


def _compare_outputs(got, want, what):
    rel = np.abs(got - want) / (np.abs(want) + 1e-3)
    med, p99 = float(np.median(rel)), float(np.percentile(rel, 99))
    obj = float(np.abs(got[..., 4] - want[..., 4]).max())
    share = float((rel <= 1e-2).mean())
    print(f"\n{what}: rel err median {med:.2e} p99 {p99:.2e}; share within 1e-2 {share:.3f}; obj abs err max {obj:.2e}")
    assert med < GATE_MED and p99 < GATE_P99 and obj < GATE_OBJ, what
    return dict(median=med, p99=p99, share_within_1e2=share, obj_abs_max=obj)


@pytest.mark.parametrize("case", TRAIN_META["cases"], ids=lambda c: c["name"])
def test_train_mode_forward_against_the_reference_vectors(case):
    """Two forward calls of the drop-in model in train() mode, then eval(): outputs against
    what the reference returned, the modules' running statistics against the oracle's."""
    from gpu_helpers import record_metric
    H, B, C, A = case["H"], case["B"], case["C"], case["A"]
    model, params, _ = build_model(H, C, A, seed=TRAIN_META["weight_seed"])
    blocks, sd = _oracle_state(C, A, TRAIN_META["weight_seed"])
    p = default_params(H, C, A)
    model.train()
    metrics = {}
    with warnings.catch_warnings():
        warnings.simplefilter("error")              # the torch-module fallback would warn
        for k, seed in enumerate(TRAIN_META["image_seeds"]):
            x_np = gi.images(seed, B, H)
            got = model(torch.from_numpy(x_np).cuda()).cpu().numpy()
            want = TRAIN_ARR[f"{case['name']}_call{k}"]
            metrics[f"call{k}"] = _compare_outputs(got, want, f"{case['name']} call {k}")
            orc.forward(sd, blocks, p, torch.from_numpy(x_np), training=True)
    assert any(key[3] for key in model._plans), "the batch-statistics plan did not run"
    msd = model.state_dict()
    worst = 0.0
    for key, dg in case["bn"].items():
        a = msd[key].double().cpu().numpy().reshape(-1)
        ref = sd[key].double().numpy().reshape(-1)
        err = float(np.linalg.norm(a - ref) / (np.linalg.norm(ref) + 1e-30))
        worst = max(worst, err)
        assert err < 8e-2, f"{key}: relative L2 error {err:.4f} vs the oracle's moved statistics"
        assert abs(np.abs(a).sum() - dg["abssum"]) < 5e-2 * dg["abssum"], key     # ... and vs the reference's
    nbt = [int(v) for k_, v in msd.items() if k_.endswith("num_batches_tracked")]
    assert len(nbt) == 72 and all(n == case["num_batches_tracked"] for n in nbt)
    model.eval()
    got = model(torch.from_numpy(gi.images(TRAIN_META["image_seeds"][0], B, H)).cuda()).cpu().numpy()
    metrics["eval_after"] = _compare_outputs(got, TRAIN_ARR[f"{case['name']}_eval_after"], f"{case['name']} eval after")
    metrics["running_stats_worst_rel_l2"] = worst
    record_metric("train_mode " + case["name"], metrics)


def test_train_mode_at_the_video_callers_size():
    """get_fps.py / video.py: one 320-pixel frame per call, model never put in eval()."""
    H, B, C, A = 320, 1, 80, 3
    model, params, _ = build_model(H, C, A, seed=3)
    blocks, sd = _oracle_state(C, A, 3)
    model.train()
    x_np = gi.images(31, B, H)
    got = model(torch.from_numpy(x_np).cuda())
    want = orc.forward(sd, blocks, default_params(H, C, A), torch.from_numpy(x_np), training=True).numpy()
    assert got.shape == (1, 6300, 85)
    _compare_outputs(got.cpu().numpy(), want, "320 / batch 1 train mode")
    # the torch modules (what the fallback runs) agree too
    with torch.no_grad():
        mod = model._forward_modules(torch.from_numpy(x_np).cuda()).cpu().numpy()
    _compare_outputs(got.cpu().numpy(), mod, "HIP path vs torch modules")


def test_train_mode_passes_on_two_streams_run_one_after_the_other():
    """A pipeline with batches in flight on several streams over a model that never called
    eval(): every pass moves the modules' running statistics with read-modify-writes, so the
    passes must not overlap.  The statistics and the outputs after four batches through the
    two-lane pipeline equal, bit for bit, those of the same four batches on one stream."""
    from yolo_amd.harness import DetectionPipeline
    from yolo_amd.utilis import filter_results
    H, B = 96, 2
    xs = [torch.from_numpy(gi.images(400 + i, B, H)).cuda() for i in range(4)]

    def run(pipelined):
        model, params, _ = build_model(H, 80, 3, seed=3)
        params = dict(params)
        params["confidence"] = 0.3
        model.train()
        outs = []
        if pipelined:
            pipe = DetectionPipeline(model, params, depth=2)
            for x in xs:
                outs += pipe.submit(x)
            outs += pipe.flush()
        else:
            for x in xs:
                outs.append(filter_results(model(x), params))
        torch.cuda.synchronize()
        assert sum(1 for key in model._plans if key[3]) == 1, "one batch-statistics plan for all streams"
        stats = {k: v.clone() for k, v in model.state_dict().items() if "bach_norm" in k}
        return [o if isinstance(o, int) else o.clone() for o in outs], stats

    want, want_stats = run(False)
    got, got_stats = run(True)
    for k, v in want_stats.items():
        assert torch.equal(got_stats[k], v), f"{k} differs between the pipelined and the serial run"
    nbt = [int(v) for k, v in got_stats.items() if k.endswith("num_batches_tracked")]
    assert nbt and all(n == 4 for n in nbt)
    assert len(got) == len(want) == 4
    for g, w in zip(got, want):
        if isinstance(w, int):
            assert isinstance(g, int)
        else:
            assert torch.equal(g.view(torch.int32), w.view(torch.int32))
    # ... and the eval() pass that follows folds the moved statistics, whatever stream moved them


def test_the_training_loss_path_is_not_offered():
    model, params, _ = build_model(64)
    x = torch.from_numpy(gi.images(12, 2, 64)).cuda()
    with pytest.raises(NotImplementedError):
        model(x, is_training=True)
