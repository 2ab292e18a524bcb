"""HIP decode, fused conv layers and the whole forward pass against the golden
vectors and the oracle.

Tolerances
  decode (fp32)            rtol 1e-5 / atol 1e-6  (torch's own sigmoid is only
                           1-ulp reproducible between strided and contiguous
                           slices, so bit equality is not available)
  single conv layer        fp32 accumulate of bf16 operands vs an fp32 torch
                           reference on the SAME bf16-rounded operands: the
                           bf16 output rounding (2^-9 relative) dominates
  whole forward (bf16)     compared with the fp32 oracle; north-star bound is
                           1e-2 relative for boxes and scores, asserted on the
                           bulk (median / 99th percentile) and printed in full
"""
import numpy as np
import pytest
import torch

import gen_inputs as gi
import yolo_oracle as orc
from helpers import assert_bit_equal, c_decode_head, c_filter_results, load_golden
from yolo_amd.cfg import default_params

from gpu_helpers import case_seed, set_knob

pytestmark = pytest.mark.gpu

META, ARR = load_golden()


# ------------------------------------------------------------------ decode
@pytest.mark.parametrize("case", META["decode_cases"], ids=lambda c: c["name"])
def test_decode_golden_nchw(case):
    from yolo_amd.yolo_v3 import decode_heads
    H, C, A, B = case["H"], case["C"], case["A"], case["B"]
    p = default_params(H, C, A)
    raws, ancs = [], []
    for k, s in enumerate((32, 16, 8)):
        raws.append(torch.from_numpy(gi.head_raw(100 + 10 * k + H, B, A, C, H // s)).cuda())
        ancs.append(p["anchors"][k * A:(k + 1) * A])
    out = decode_heads(raws, ancs, H, C).cpu().numpy()
    np.testing.assert_allclose(out, ARR[case["name"]], rtol=1e-5, atol=1e-6)


def test_decode_overflow_and_predict_method():
    from gpu_helpers import build_model
    model, params, _ = build_model(64)
    raw = gi.head_raw(7, 1, 3, 80, 2)
    raw[0, 2, 0, 0] = 100.0
    raw[0, 85 + 3, 1, 1] = 95.0
    layer = model.module_list[82]
    got = model.predict(torch.from_numpy(raw).cuda(), layer, 1).cpu().numpy()
    want = ARR["dec_overflow"]
    assert np.array_equal(np.isinf(got), np.isinf(want)) and np.isinf(got).sum() == 2
    fin = np.isfinite(want)
    np.testing.assert_allclose(got[fin], want[fin], rtol=1e-5, atol=1e-6)


def test_decode_large_and_saturating_logits():
    """The decode uses the hardware exp2 / reciprocal, not expf and an IEEE division: its
    contract is 1e-5 relative, and the error of exp(x) grows with |x|.  Logits of 10-20 in
    every column, the whole normal range of exp (|x| up to 87) and saturating values stay
    inside the contract; overflow gives inf, underflow 0, as in the reference."""
    from yolo_amd.yolo_v3 import decode_heads
    H, C, A, G = 416, 80, 3, 13
    p = default_params(H, C, A)
    r = gi.rs(4242)
    raw = (r.uniform(10.0, 20.0, (2, A * (5 + C), G, G)) * r.choice([-1.0, 1.0], (2, A * (5 + C), G, G)))
    raw = raw.astype(np.float32)
    special = np.array([30.0, -30.0, 60.0, -60.0, 87.0, -87.0, 88.5, -88.5, 100.0, -100.0, 1000.0, -1000.0,
                        0.0, -0.0, 1e-30, 16.635532, -16.635532], dtype=np.float32)
    for ch in range(A * (5 + C)):                    # every column type sees every special value
        raw[1, ch].reshape(-1)[:special.size] = np.roll(special, ch)
    anchors = p["anchors"][:A]
    got = decode_heads([torch.from_numpy(raw).cuda()], [anchors], H, C).cpu().numpy()
    want = c_decode_head(raw, anchors, H, C)
    assert np.array_equal(np.isinf(got), np.isinf(want)) and np.isinf(want).sum() > 0
    assert not np.isnan(got).any()
    fin = np.isfinite(want)
    np.testing.assert_allclose(got[fin], want[fin], rtol=1e-5, atol=1e-6)


def test_decode_full_size_vs_c_oracle():
    from yolo_amd.yolo_v3 import decode_heads
    p = default_params(416)
    raws, ancs, want = [], [], []
    for k, s in enumerate((32, 16, 8)):
        r = gi.head_raw(300 + k, 2, 3, 80, 416 // s)
        raws.append(torch.from_numpy(r).cuda())
        ancs.append(p["anchors"][k * 3:(k + 1) * 3])
        want.append(c_decode_head(r, ancs[-1], 416, 80))
    out = decode_heads(raws, ancs, 416, 80).cpu().numpy()
    assert out.shape == (2, 10647, 85)
    np.testing.assert_allclose(out, np.concatenate(want, 1), rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ single conv layers
CONV_CASES = [
    # name, B, H, c0, cout, k, stride, leaky, residual, c1, up0, fp32_out
    ("k3s2_c32_n64", 2, 26, 32, 64, 3, 2, True, False, 0, 0, False),      # BK=32 BN=64
    ("k1_c64_n32", 2, 26, 64, 32, 1, 1, True, False, 0, 0, False),        # BK=64 BN=32
    ("k3_c32_n64_res", 3, 13, 32, 64, 3, 1, True, True, 0, 0, False),     # BK=32 BN=64 + residual
    ("k3_c32_n32", 1, 20, 32, 32, 3, 1, True, False, 0, 0, False),        # BK=32 BN=32
    ("k3s2_c64_n128", 2, 26, 64, 128, 3, 2, True, False, 0, 0, False),    # few tiles: 64-channel tiles
    ("k1_c128_n64", 3, 13, 128, 64, 1, 1, True, False, 0, 0, False),      # BN=64
    ("k3_c96_n128", 1, 15, 96, 128, 3, 1, True, True, 0, 0, False),       # BK=32 + residual (BN=128 cases: BN128_CASES)
    ("k3_c128_n256_res", 2, 13, 128, 256, 3, 1, True, True, 0, 0, False), # several channel tiles
    ("head_c256_n255", 2, 13, 256, 255, 1, 1, False, False, 0, 0, True),  # detection conv, fp32 out
    ("head_c128_n9", 1, 10, 128, 9, 1, 1, False, False, 0, 0, True),      # one-anchor head
    ("route_up_c64_c128", 2, 12, 64, 64, 1, 1, True, False, 128, 1, False),   # upsample + concat
    ("concat_k3_c64_c64", 1, 9, 64, 128, 3, 1, True, False, 64, 0, False),    # concat into 3x3
    ("big_m_tail", 5, 19, 64, 64, 3, 1, True, False, 0, 0, False),        # M = 1805, ragged tiles
    ("thin_c32_n64_plain", 2, 37, 32, 64, 3, 1, False, False, 0, 0, False),   # thin kernel, ragged 8x16 tiles, linear
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: c[0])
def test_conv_layer(case):
    from gpu_helpers import conv_reference, run_conv
    name, B, H, c0, cout, k, stride, leaky, use_res, c1, up0, fp32_out = case
    r = gi.rs(case_seed(name))
    H0 = H >> up0
    x0 = torch.from_numpy(r.standard_normal((B, H0, H0, c0)).astype(np.float32)).cuda().to(torch.bfloat16)
    x1 = None
    if c1:
        x1 = torch.from_numpy(r.standard_normal((B, H, H, c1)).astype(np.float32)).cuda().to(torch.bfloat16)
    cin = c0 + c1
    w = torch.from_numpy((r.standard_normal((cout, cin, k, k)) * np.sqrt(2.0 / (cin * k * k))).astype(np.float32))
    bias = torch.from_numpy(r.normal(0, 0.3, cout).astype(np.float32))
    pad = k // 2
    Ho = (H + 2 * pad - k) // stride + 1
    res = None
    if use_res:
        res = torch.from_numpy(r.standard_normal((B, Ho, Ho, cout)).astype(np.float32)).cuda().to(torch.bfloat16)
    got = run_conv(x0, w, bias, stride=stride, leaky=leaky, x1=x1, up0=up0, residual=res, fp32_out=fp32_out)
    want = conv_reference(x0, w, bias, stride=stride, leaky=leaky, x1=x1, up0=up0, residual=res)
    got = got.float().cpu()
    assert got.shape == want.shape
    assert torch.isfinite(got).all(), "output holds NaN/inf (unwritten elements?)"
    if fp32_out:
        torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-4)
    else:
        torch.testing.assert_close(got, want, rtol=1.0 / 128, atol=1.0 / 128)


# The im2col kernel takes 128-channel tiles only when ceil(M/128) * cout_pad/128 >= 384
# (large batches): these are the shapes of the 416 / batch-32 benchmark that do, plus
# one per instantiated (K step, ring depth) pair.  Every case asserts the kernel and
# tiling the library reports for it, so a change of the dispatch heuristic can never
# silently leave an instantiation without a parity check.
BN128_CASES = [
    # name, B, H, c0, cout, k, stride, residual, knobs, expected (bn, bk, act_ksteps, lds_stages)
    ("bn128_k1_c256_n128_h52", 32, 52, 256, 128, 1, 1, False, {}, (128, 32, 2, 2)),          # 10 layers of the benchmark
    ("bn128_k3s2_c256_n512_h52", 32, 52, 256, 512, 3, 2, False, {}, (128, 32, 2, 2)),        # down-sampling layer
    ("bn128_k3s2_c64_n128_h208", 8, 208, 64, 128, 3, 2, False, {}, (128, 32, 2, 2)),
    ("bn128_k1_c256_n256_ragged_res", 20, 50, 256, 256, 1, 1, True, {}, (128, 32, 2, 2)),    # M = 50000: ragged last tile
    ("bn128_k1_c512_n256_h52", 16, 52, 512, 256, 1, 1, False, {}, (128, 64, 1, 2)),          # K step 64
    ("bn128_bk64_ring3", 16, 52, 512, 256, 1, 1, True, {"YOLO_AMD_CONV_NSTAGE": "3"}, (128, 64, 1, 3)),
    ("bn128_bk64_ring4", 16, 52, 512, 256, 1, 1, False, {"YOLO_AMD_CONV_NSTAGE": "4"}, (128, 64, 1, 4)),
    ("bn128_bk32_amul1", 32, 52, 256, 128, 1, 1, False, {"YOLO_AMD_CONV_AMUL": "1"}, (128, 32, 1, 2)),
    ("bn128_bk32_ring3", 32, 52, 256, 128, 1, 1, True, {"YOLO_AMD_CONV_NSTAGE": "3"}, (128, 32, 1, 3)),
    ("bn128_bk32_ring4", 32, 52, 256, 128, 1, 1, False, {"YOLO_AMD_CONV_NSTAGE": "4"}, (128, 32, 1, 4)),
]


@pytest.mark.parametrize("case", BN128_CASES, ids=lambda c: c[0])
def test_conv_layer_128_channel_tiles(case, monkeypatch):
    from gpu_helpers import conv_reference, run_conv
    name, B, H, c0, cout, k, stride, use_res, knobs, expect = case
    for kn, v in knobs.items():
        set_knob(monkeypatch, kn, v)
    g = torch.Generator(device="cuda").manual_seed(case_seed(name))
    x0 = torch.randn((B, H, H, c0), generator=g, device="cuda").to(torch.bfloat16)
    r = gi.rs(case_seed(name))
    w = torch.from_numpy((r.standard_normal((cout, c0, k, k)) * np.sqrt(2.0 / (c0 * k * k))).astype(np.float32))
    bias = torch.from_numpy(r.normal(0, 0.3, cout).astype(np.float32))
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    res = None
    if use_res:
        res = torch.randn((B, Ho, Ho, cout), generator=g, device="cuda").to(torch.bfloat16)
    info = {}
    got = run_conv(x0, w, bias, stride=stride, leaky=True, residual=res, info=info).float().cpu()
    ch = info["choice"]
    assert ch["kernel"] == "igemm" and (ch["bn"], ch["bk"], ch["act_ksteps"], ch["lds_stages"]) == expect and \
        ch["k_slices"] == 1, f"this shape was meant for the {expect} im2col kernel, the library picked {ch}"
    want = conv_reference(x0, w, bias, stride=stride, leaky=True, residual=res)
    assert torch.isfinite(got).all(), "output holds NaN/inf (unwritten elements?)"
    torch.testing.assert_close(got, want, rtol=1.0 / 128, atol=1.0 / 128)


def test_a_detection_layer_with_decode_never_takes_the_patch_or_thin_kernel():
    """A 3x3 stride-1 detection convolution on a 13*2^n map with cout_pad % 128 == 0
    fits the patch kernel's shape test; that kernel has no decode epilogue, so the
    library must route it to the im2col kernel -- and the rows must be decoded."""
    from yolo_amd import _lib
    from yolo_amd._lib import ConvDesc, HeadDesc, check, lib
    from yolo_amd.engine import pack_conv_weights
    from gpu_helpers import conv_reference
    from helpers import c_decode_head
    B, G, cin, A, C = 32, 26, 64, 3, 80
    D, cout = 5 + C, 3 * 85
    r = gi.rs(case_seed("decode_3x3"))
    x0 = torch.from_numpy(r.standard_normal((B, G, G, cin)).astype(np.float32)).cuda().to(torch.bfloat16)
    w = torch.from_numpy((r.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (cin * 9))).astype(np.float32))
    bias = torch.from_numpy(r.normal(0, 0.3, cout).astype(np.float32))
    wp = pack_conv_weights(w.permute(0, 2, 3, 1).reshape(cout, 9 * cin).cuda(), 256)
    bp = torch.zeros(256, dtype=torch.float32, device="cuda")
    bp[:cout] = bias.cuda()
    N = G * G * A
    out = torch.full((B, N, D), float("nan"), dtype=torch.float32, device="cuda")
    zero = torch.zeros(1024, dtype=torch.uint8, device="cuda")
    anchors = np.array([[30.0, 61.0], [62.0, 45.0], [59.0, 119.0]])
    hd = HeadDesc()
    hd.G, hd.layout, hd.row_offset, hd.stride = G, _lib.LAYOUT_NHWC, 0, 16.0
    for a in range(A):
        hd.anchor_w[a] = np.float32(anchors[a][0] / 16)
        hd.anchor_h[a] = np.float32(anchors[a][1] / 16)
    d = (ConvDesc * 1)()
    d[0].src0, d[0].weight, d[0].bias, d[0].dst = x0.data_ptr(), wp.data_ptr(), bp.data_ptr(), out.data_ptr()
    d[0].c0, d[0].B, d[0].H, d[0].W = cin, B, G, G
    d[0].cout, d[0].cout_pad, d[0].ksize, d[0].stride, d[0].k_pad = cout, 256, 3, 1, 9 * cin
    d[0].leaky, d[0].dst_fp32 = 0, 1
    import ctypes
    assert _lib.conv_choice(d[0])["kernel"] == "patch"            # the shape alone would take it ...
    d[0].decode, d[0].decode_rows, d[0].decode_anchors = ctypes.pointer(hd), N, A
    assert _lib.conv_choice(d[0])["kernel"] == "igemm"            # ... a layer that decodes does not
    check(lib.yolo_amd_conv_forward(d, 1, zero.data_ptr(), _lib.current_stream_ptr(x0.device)), "conv")
    torch.cuda.synchronize()
    raw = conv_reference(x0, w, bias, leaky=False)                 # (B, G, G, A*D) fp32
    want = c_decode_head(raw.permute(0, 3, 1, 2).contiguous().numpy(), anchors, 416, C)
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("B,H,use_res", [(2, 40, True), (1, 21, False), (33, 16, True)])
def test_thin_kernel_gives_the_im2col_kernels_bits(B, H, use_res, monkeypatch):
    """3x3 32 -> 64 layers run in the persistent thin kernel; its K-steps are in the
    im2col kernel's order, so the two must agree bit for bit."""
    from gpu_helpers import run_conv
    r = gi.rs(900 + H)
    x0 = torch.from_numpy(r.standard_normal((B, H, H, 32)).astype(np.float32)).cuda().to(torch.bfloat16)
    w = torch.from_numpy((r.standard_normal((64, 32, 3, 3)) * np.sqrt(2.0 / 288)).astype(np.float32))
    bias = torch.from_numpy(r.normal(0, 0.3, 64).astype(np.float32))
    res = None
    if use_res:
        res = torch.from_numpy(r.standard_normal((B, H, H, 64)).astype(np.float32)).cuda().to(torch.bfloat16)
    thin = run_conv(x0, w, bias, leaky=True, residual=res)
    set_knob(monkeypatch, "YOLO_AMD_THIN", "0")
    generic = run_conv(x0, w, bias, leaky=True, residual=res)
    assert torch.equal(thin.view(torch.int16), generic.view(torch.int16))


# stride-1 3x3 layers on 13*2^n maps can take the patch-reuse kernel
@pytest.mark.parametrize("B,H,W", [(2, 37, 37), (1, 52, 40), (3, 8, 16), (1, 5, 3)])
def test_first_residual_block_in_one_launch_gives_the_two_launches_bits(B, H, W):
    """1x1 64 -> 32, 3x3 32 -> 64, shortcut from the block's input: descriptor 0 with
    fuse_next runs the pair as one launch (conv_block64_kernel); the bits must be those of
    the two separate launches (im2col kernel, then thin kernel), and close to the fp32
    reference on the same bf16 operands.  Ragged tiles, tiles that hang over both edges."""
    from gpu_helpers import conv_reference, run_conv
    from yolo_amd import _lib
    from yolo_amd._lib import ConvDesc, check, lib
    from yolo_amd.engine import pack_conv_weights
    r = gi.rs(case_seed(f"block64_{B}_{H}_{W}"))
    dev = torch.device("cuda")
    x = torch.from_numpy(r.standard_normal((B, H, W, 64)).astype(np.float32)).to(dev).to(torch.bfloat16)
    w1 = torch.from_numpy((r.standard_normal((32, 64, 1, 1)) * np.sqrt(2.0 / 64)).astype(np.float32))
    b1 = torch.from_numpy(r.normal(0, 0.3, 32).astype(np.float32))
    w3 = torch.from_numpy((r.standard_normal((64, 32, 3, 3)) * np.sqrt(2.0 / 288)).astype(np.float32))
    b3 = torch.from_numpy(r.normal(0, 0.3, 64).astype(np.float32))
    # run_conv works on square maps only through its H = W assumption? it takes the shape from x0
    info1, info3 = {}, {}
    y1 = run_conv(x, w1, b1, leaky=True, info=info1)
    y2 = run_conv(y1, w3, b3, leaky=True, residual=x, info=info3)
    assert info1["choice"]["kernel"] == "igemm" and info3["choice"]["kernel"] == "thin"

    wp1 = pack_conv_weights(w1.permute(0, 2, 3, 1).reshape(32, 64).to(dev), 32)
    wp3 = pack_conv_weights(w3.permute(0, 2, 3, 1).reshape(64, 288).to(dev), 64)
    bp1, bp3 = b1.to(dev), b3.to(dev)
    mid = torch.full((B, H, W, 32), float("nan"), dtype=torch.bfloat16, device=dev)
    out = torch.full((B, H, W, 64), float("nan"), dtype=torch.bfloat16, device=dev)
    zero = torch.zeros(1024, dtype=torch.uint8, device=dev)
    d = (ConvDesc * 2)()
    d[0].src0, d[0].weight, d[0].bias, d[0].dst = x.data_ptr(), wp1.data_ptr(), bp1.data_ptr(), mid.data_ptr()
    d[0].c0, d[0].c1, d[0].up0, d[0].B, d[0].H, d[0].W = 64, 0, 0, B, H, W
    d[0].cout, d[0].cout_pad, d[0].ksize, d[0].stride, d[0].k_pad, d[0].leaky = 32, 32, 1, 1, 64, 1
    d[0].fuse_next = 1
    d[1].src0, d[1].weight, d[1].bias, d[1].dst = mid.data_ptr(), wp3.data_ptr(), bp3.data_ptr(), out.data_ptr()
    d[1].residual = x.data_ptr()
    d[1].c0, d[1].c1, d[1].up0, d[1].B, d[1].H, d[1].W = 32, 0, 0, B, H, W
    d[1].cout, d[1].cout_pad, d[1].ksize, d[1].stride, d[1].k_pad, d[1].leaky = 64, 64, 3, 1, 288, 1
    assert _lib.conv_choice(d[0])["kernel"] == "block"
    for _ in range(2):                       # twice: the second launch must write the same bits
        out.fill_(float("nan"))
        check(lib.yolo_amd_conv_forward(d, 2, zero.data_ptr(), _lib.current_stream_ptr(dev)), "fused block")
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int16), y2.view(torch.int16)), "fused block differs from the two launches"
    assert bool(torch.isnan(mid.float()).all()), "the 1x1 layer's output was written although the pair is fused"
    want = conv_reference(conv_reference(x, w1, b1).to(torch.bfloat16).cuda(), w3, b3, residual=x)
    torch.testing.assert_close(out.float().cpu(), want, rtol=1.0 / 128, atol=1.0 / 128)
    # a pair the kernel is not built for is refused, not mis-run
    d[1].residual = mid.data_ptr()
    assert lib.yolo_amd_conv_forward(d, 2, zero.data_ptr(), _lib.current_stream_ptr(dev)) < 0
    assert b"shortcut operand" in lib.yolo_amd_last_error()


def test_a_plan_with_the_fused_block_gives_the_default_plans_bits(monkeypatch):
    """YOLO_AMD_BLOCK=1: the plan runs the network's first residual block as one launch (off
    by default: measured slower).  Same prediction tensor, bit for bit, as the default plan."""
    from gpu_helpers import build_model
    x = torch.from_numpy(gi.images(33, 3, 160)).cuda()
    model, _, _ = build_model(160)
    want = model(x).clone()
    assert not model._plan(3, 160).fused_block
    monkeypatch.setenv("YOLO_AMD_BLOCK", "1")
    model.refold()                                     # new plans
    got = model(x)
    plan = model._plan(3, 160)
    assert plan.fused_block and sum(int(d.fuse_next) for d in plan.descs) == 2      # the stem and the block
    assert torch.equal(got, want)


PATCH_CASES = [
    # name, B, H, cin, cout, leaky, residual, fp32_out
    ("p3_c32_n128_short_k", 1, 13, 32, 128, True, False, False),
    ("p3_c64_n128_b3", 3, 26, 64, 128, True, True, False),
    ("p3_c96_n256_one_tile", 1, 13, 96, 256, True, True, False),
    ("p3_c256_n512_h26", 2, 26, 256, 512, True, True, False),
    ("p3_c128_n255_fp32", 2, 13, 128, 255, False, False, True),
    ("p3_c64_n128_h52", 1, 52, 64, 128, True, False, False),
    ("p3_c64_n768_h26_b5", 5, 26, 64, 768, True, True, False),     # three 256-channel tiles, or six of 128
    ("p3_c64_n192_partial_channel_tile", 2, 13, 64, 192, True, True, False),   # the second tile holds 64 real channels
    ("p3_c32_n200_partial_chunk", 1, 26, 32, 200, True, True, False),          # ... 72: a wave's 32 channels end mid-way
    # 19 * 2^n maps (608-pixel inputs): 19 x 10 tiles, the last tile of a row hangs over the edge
    ("p3_19x10_c64_n128_h19", 2, 19, 64, 128, True, True, False),
    ("p3_19x10_c96_n256_h38", 3, 38, 96, 256, True, True, False),
    ("p3_19x10_c32_n255_h76_fp32", 1, 76, 32, 255, False, False, True),
]


@pytest.mark.parametrize("stg", ["1", "3", "3/256", "9", "9/256", "9/flat", "9/flat/6"],
                         ids=["tap-stages", "row-stages", "row-stages-256ch", "private-weight-rings",
                              "private-weight-rings-256ch", "flat-patch", "flat-patch-six-and-five-groups"])
@pytest.mark.parametrize("case", PATCH_CASES, ids=lambda c: c[0])
def test_conv3x3_patch_kernel(case, stg, monkeypatch):
    from gpu_helpers import conv_reference, run_conv
    name, B, H, cin, cout, leaky, use_res, fp32_out = case
    set_knob(monkeypatch, "YOLO_AMD_CONV3", "1")
    wide = stg.endswith("/256")
    if wide and (cout + 127) // 128 * 128 % 256:
        pytest.skip("the 256-channel tile needs the padded channel count to be a multiple of 256")
    flat = "1" if stg.startswith("9/flat") else "0"            # the default form; "9" is the same rings on rotated patch rows
    if stg == "9/flat/6":
        if H % 13:
            pytest.skip("the 19 x 10 tile has one split of its twelve pixel groups")
        set_knob(monkeypatch, "YOLO_AMD_CONV3_SPLIT", "6")    # waves 0-3 six pixel groups, waves 4-7 five (default: seven and four)
    if H % 13 and stg not in ("1", "9") and flat == "0":
        pytest.skip("19 x 10 tiles exist with tap-deep weight stages and with private weight rings")
    set_knob(monkeypatch, "YOLO_AMD_CONV3_STG", stg[0])
    set_knob(monkeypatch, "YOLO_AMD_CONV3_TC", "4" if wide else "2")
    set_knob(monkeypatch, "YOLO_AMD_CONV3_FLAT", flat)
    r = gi.rs(case_seed(name))
    x0 = torch.from_numpy(r.standard_normal((B, H, H, cin)).astype(np.float32)).cuda().to(torch.bfloat16)
    w = torch.from_numpy((r.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (cin * 9))).astype(np.float32))
    bias = torch.from_numpy(r.normal(0, 0.3, cout).astype(np.float32))
    res = None
    if use_res:
        res = torch.from_numpy(r.standard_normal((B, H, H, cout)).astype(np.float32)).cuda().to(torch.bfloat16)
    info = {}
    got = run_conv(x0, w, bias, leaky=leaky, residual=res, fp32_out=fp32_out, info=info).float().cpu()
    ch = info["choice"]
    assert ch["kernel"] == "patch" and ch["bn"] == (256 if wide else 128) and ch["taps_per_stage"] == int(stg[0]), ch
    assert ch["patch_flat"] == int(flat), ch
    want = conv_reference(x0, w, bias, leaky=leaky, residual=res)
    assert torch.isfinite(got).all(), "output holds NaN/inf (unwritten elements?)"
    if flat != "0":
        # the flat-patch kernel adds the same products in the same order as the rotated-patch kernel with
        # private weight rings: the same bits
        set_knob(monkeypatch, "YOLO_AMD_CONV3_FLAT", "0")
        rotated = run_conv(x0, w, bias, leaky=leaky, residual=res, fp32_out=fp32_out).float().cpu()
        assert torch.equal(got, rotated), "flat-patch kernel differs from the rotated-patch kernel"
    # and the generic im2col kernel must agree with it
    set_knob(monkeypatch, "YOLO_AMD_CONV3", "0")
    generic = run_conv(x0, w, bias, leaky=leaky, residual=res, fp32_out=fp32_out).float().cpu()
    if fp32_out:
        torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-4)
        torch.testing.assert_close(got, generic, rtol=2e-4, atol=2e-4)
    else:
        torch.testing.assert_close(got, want, rtol=1.0 / 128, atol=1.0 / 128)
        torch.testing.assert_close(got, generic, rtol=1.0 / 128, atol=1.0 / 128)


def test_first_layer():
    from yolo_amd import _lib
    from yolo_amd._lib import ConvDesc, lib, check
    import torch.nn.functional as F
    r = gi.rs(5)
    B, H = 2, 37
    x = torch.from_numpy(gi.images(5, B, H)).cuda()
    w = torch.from_numpy((r.standard_normal((32, 3, 3, 3)) * 0.27).astype(np.float32))
    bias = torch.from_numpy(r.normal(0, 0.3, 32).astype(np.float32))
    wk = w.permute(0, 2, 3, 1).reshape(32, 27).t().contiguous().cuda()
    bk = bias.cuda()
    out = torch.full((B, H, H, 32), float("nan"), dtype=torch.bfloat16, device="cuda")
    d = (ConvDesc * 1)()
    d[0].src0, d[0].weight, d[0].bias, d[0].dst = x.data_ptr(), wk.data_ptr(), bk.data_ptr(), out.data_ptr()
    d[0].c0, d[0].c1, d[0].up0, d[0].B, d[0].H, d[0].W = 3, 0, 0, B, H, H
    d[0].cout, d[0].cout_pad, d[0].ksize, d[0].stride, d[0].k_pad = 32, 32, 3, 1, 27
    d[0].leaky, d[0].dst_fp32, d[0].first_layer = 1, 0, 1
    check(lib.yolo_amd_conv_forward(d, 1, None, _lib.current_stream_ptr(x.device)), "first conv")
    torch.cuda.synchronize()
    want = F.leaky_relu(F.conv2d(x.cpu(), w, bias, padding=1), 0.1).permute(0, 2, 3, 1)
    torch.testing.assert_close(out.float().cpu(), want, rtol=1.0 / 128, atol=1.0 / 128)


@pytest.mark.parametrize("B,H,W", [(2, 37, 37), (1, 64, 48), (3, 9, 33), (1, 416, 416)])
def test_stem_in_one_launch_gives_the_separate_launches_bits(B, H, W):
    """First layer (3 -> 32), stride-2 layer (32 -> 64), 1x1 layer (64 -> 32): fuse_next = 1 on
    the first descriptor runs the first two as one launch, fuse_next = 2 all three; both the
    64-channel and the 32-channel output must be the bits of three separate launches.  Odd
    sizes (tiles over both edges) and the benchmark's 416 x 416."""
    from yolo_amd import _lib
    from yolo_amd._lib import ConvDesc, lib, check
    from yolo_amd.engine import pack_conv_weights
    r = gi.rs(case_seed(f"stem_{B}_{H}_{W}"))
    dev = torch.device("cuda")
    x = torch.from_numpy(r.uniform(0, 1, (B, 3, H, W)).astype(np.float32)).to(dev)
    w0 = torch.from_numpy((r.standard_normal((32, 3, 3, 3)) * 0.27).astype(np.float32))
    w1 = torch.from_numpy((r.standard_normal((64, 32, 3, 3)) * np.sqrt(2.0 / 288)).astype(np.float32))
    w2 = torch.from_numpy((r.standard_normal((32, 64, 1, 1)) * np.sqrt(2.0 / 64)).astype(np.float32))
    wk0 = w0.permute(0, 2, 3, 1).reshape(32, 27).t().contiguous().to(dev)
    wk1 = pack_conv_weights(w1.permute(0, 2, 3, 1).reshape(64, 288).to(dev), 64)
    wk2 = pack_conv_weights(w2.permute(0, 2, 3, 1).reshape(32, 64).to(dev), 32)
    b0, b1, b2 = (torch.from_numpy(r.normal(0, 0.3, c).astype(np.float32)).to(dev) for c in (32, 64, 32))
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    zero = torch.zeros(1024, dtype=torch.uint8, device=dev)
    stream = _lib.current_stream_ptr(dev)

    def run(fuse):
        l0 = torch.full((B, H, W, 32), float("nan"), dtype=torch.bfloat16, device=dev)
        l1 = torch.full((B, Ho, Wo, 64), float("nan"), dtype=torch.bfloat16, device=dev)
        l2 = torch.full((B, Ho, Wo, 32), float("nan"), dtype=torch.bfloat16, device=dev)
        d = (ConvDesc * 3)()
        d[0].src0, d[0].weight, d[0].bias, d[0].dst = x.data_ptr(), wk0.data_ptr(), b0.data_ptr(), l0.data_ptr()
        d[0].c0, d[0].B, d[0].H, d[0].W = 3, B, H, W
        d[0].cout, d[0].cout_pad, d[0].ksize, d[0].stride, d[0].k_pad = 32, 32, 3, 1, 27
        d[0].leaky, d[0].first_layer, d[0].fuse_next = 1, 1, fuse
        d[1].src0, d[1].weight, d[1].bias, d[1].dst = l0.data_ptr(), wk1.data_ptr(), b1.data_ptr(), l1.data_ptr()
        d[1].c0, d[1].B, d[1].H, d[1].W = 32, B, H, W
        d[1].cout, d[1].cout_pad, d[1].ksize, d[1].stride, d[1].k_pad, d[1].leaky = 64, 64, 3, 2, 288, 1
        d[2].src0, d[2].weight, d[2].bias, d[2].dst = l1.data_ptr(), wk2.data_ptr(), b2.data_ptr(), l2.data_ptr()
        d[2].c0, d[2].B, d[2].H, d[2].W = 64, B, Ho, Wo
        d[2].cout, d[2].cout_pad, d[2].ksize, d[2].stride, d[2].k_pad, d[2].leaky = 32, 32, 1, 1, 64, 1
        check(lib.yolo_amd_conv_forward(d, 3, zero.data_ptr(), stream), f"stem, fuse_next {fuse}")
        torch.cuda.synchronize()
        if fuse:
            assert bool(torch.isnan(l0.float()).all()), "the first layer's output was written although it is fused"
        return l1, l2, d

    want1, want2, _ = run(0)
    assert not bool(torch.isnan(want2.float()).any())
    for fuse in (1, 2):
        got1, got2, d = run(fuse)
        assert torch.equal(got1.view(torch.int16), want1.view(torch.int16)), f"fuse_next {fuse}: 64-channel output"
        assert torch.equal(got2.view(torch.int16), want2.view(torch.int16)), f"fuse_next {fuse}: 32-channel output"
    # a third layer the kernel is not built for is refused, not mis-run
    d[2].leaky, d[2].stride = 1, 2
    assert lib.yolo_amd_conv_forward(d, 3, zero.data_ptr(), stream) < 0
    assert b"third layer" in lib.yolo_amd_last_error()
    d[2].stride = 1
    assert lib.yolo_amd_conv_forward(d, 2, zero.data_ptr(), stream) < 0      # reaches past the array


@pytest.mark.parametrize("B,H,W,stride,res,kind", [
    (2, 52, 52, 2, False, "igemm2"), (1, 37, 21, 2, False, "igemm2"), (3, 9, 16, 1, True, "igemm2"),
    (2, 104, 104, 2, False, "igemm2"), (1, 26, 26, 2, True, "igemm1"),
    (4, 104, 104, 1, True, "flat"), (8, 52, 52, 1, False, "flat"), (2, 152, 152, 1, True, "flat"),
    (3, 104, 91, 1, True, "flat"),
    # ragged shapes from a fixed seed: tiles that hang over the map on both sides, pixel counts that are not
    # multiples of the 128-pixel tile, both strides, with and without a shortcut
    *[(int(b), int(h), int(w), int(st), bool(rs), "igemm2") for b, h, w, st, rs in
      np.random.RandomState(20260112).randint([1, 5, 5, 1, 0], [4, 70, 70, 3, 2], size=(10, 5))]])
def test_a_1x1_layer_in_the_epilogue_of_the_3x3_layer_in_front_gives_the_two_launches_bits(B, H, W, stride, res, kind,
                                                                                           monkeypatch):
    """3x3 64 -> 128 (stride 2, or stride 1 with a shortcut) with fuse_next, then 1x1 128 -> 64
    on its output: one launch where the 3x3 layer runs the 128-channel-tile im2col kernel with the
    two-step activation tile ("igemm2") or the patch-reuse kernel's flat form ("flat": 13 x 13 and
    19 x 10 tiles, with and without a shortcut), two where it does not (the 26 x 26 case is made to
    take the one-step activation tile); both outputs must be the bits of two separate launches."""
    from yolo_amd import _lib
    from yolo_amd._lib import ConvDesc, lib, check
    from yolo_amd.engine import pack_conv_weights
    set_knob(monkeypatch, "YOLO_AMD_CONV_BN128_MIN", "0")      # small cases keep the 128-channel tile
    set_knob(monkeypatch, "YOLO_AMD_CONV_BN64_MIN", "0")
    if H == 26:
        set_knob(monkeypatch, "YOLO_AMD_CONV_AMUL", "1")
    r = gi.rs(case_seed(f"tail_{B}_{H}_{W}_{stride}"))
    dev = torch.device("cuda")
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    x = torch.from_numpy(r.standard_normal((B, H, W, 64)).astype(np.float32)).to(dev).to(torch.bfloat16)
    rs_ = torch.from_numpy(r.standard_normal((B, Ho, Wo, 128)).astype(np.float32)).to(dev).to(torch.bfloat16)
    w3 = torch.from_numpy((r.standard_normal((128, 64, 3, 3)) * np.sqrt(2.0 / 576)).astype(np.float32))
    w1 = torch.from_numpy((r.standard_normal((64, 128, 1, 1)) * np.sqrt(2.0 / 128)).astype(np.float32))
    wk3 = pack_conv_weights(w3.permute(0, 2, 3, 1).reshape(128, 576).to(dev), 128)
    wk1 = pack_conv_weights(w1.permute(0, 2, 3, 1).reshape(64, 128).to(dev), 64)
    b3, b1 = (torch.from_numpy(r.normal(0, 0.3, c).astype(np.float32)).to(dev) for c in (128, 64))
    zero = torch.zeros(1024, dtype=torch.uint8, device=dev)
    stream = _lib.current_stream_ptr(dev)

    def run(fuse):
        y = torch.full((B, Ho, Wo, 128), float("nan"), dtype=torch.bfloat16, device=dev)
        z = torch.full((B, Ho, Wo, 64), float("nan"), dtype=torch.bfloat16, device=dev)
        d = (ConvDesc * 2)()
        d[0].src0, d[0].weight, d[0].bias, d[0].dst = x.data_ptr(), wk3.data_ptr(), b3.data_ptr(), y.data_ptr()
        d[0].residual = rs_.data_ptr() if res else None
        d[0].c0, d[0].B, d[0].H, d[0].W = 64, B, H, W
        d[0].cout, d[0].cout_pad, d[0].ksize, d[0].stride, d[0].k_pad, d[0].leaky = 128, 128, 3, stride, 576, 1
        d[0].fuse_next = fuse
        d[1].src0, d[1].weight, d[1].bias, d[1].dst = y.data_ptr(), wk1.data_ptr(), b1.data_ptr(), z.data_ptr()
        d[1].c0, d[1].B, d[1].H, d[1].W = 128, B, Ho, Wo
        d[1].cout, d[1].cout_pad, d[1].ksize, d[1].stride, d[1].k_pad, d[1].leaky = 64, 64, 1, 1, 128, 1
        check(lib.yolo_amd_conv_forward(d, 2, zero.data_ptr(), stream), f"3x3 + 1x1, fuse_next {fuse}")
        torch.cuda.synchronize()
        return y, z, d

    y0, z0, d = run(0)
    assert not bool(torch.isnan(y0.float()).any()) and not bool(torch.isnan(z0.float()).any())
    ch = _lib.conv_choice(d[0])
    print(f"\n3x3 layer: {ch}")
    if kind == "flat":
        assert ch["kernel"] == "patch" and ch["patch_flat"] == 1 and ch["bn"] == 128, "meant to cover the flat patch-reuse kernel"
    else:
        assert ch["kernel"] == "igemm" and ch["bn"] == 128
        assert ch["act_ksteps"] == (1 if kind == "igemm1" else 2), "igemm1 covers the two-launch form, igemm2 the one-launch form"
    for _ in range(2):
        y1, z1, d = run(1)
        assert torch.equal(y1.view(torch.int16), y0.view(torch.int16)), "3x3 layer's output"
        assert torch.equal(z1.view(torch.int16), z0.view(torch.int16)), "1x1 layer's output"
    d[1].stride = 2
    assert lib.yolo_amd_conv_forward(d, 2, zero.data_ptr(), stream) < 0
    assert b"fuse_next on a 3x3 layer" in lib.yolo_amd_last_error()


# ------------------------------------------------------------------ whole forward
def _rel_err(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / (np.linalg.norm(b) + 1e-30))


@pytest.mark.parametrize("case", META["forward_cases"], ids=lambda c: c["name"])
def test_forward_layers_and_output(case):
    from gpu_helpers import build_model, oracle_forward
    H, B, C, A = case["H"], case["B"], case["C"], case["A"]
    model, params, sd = build_model(H, C, A, seed=case["weight_seed"])
    x_np = gi.images(case["image_seed"], B, H)
    det = model(torch.from_numpy(x_np).cuda())
    torch.cuda.synchronize()
    assert list(det.shape) == case["out_shape"] and det.dtype == torch.float32 and det.is_cuda
    want, outs = oracle_forward(sd, H, C, A, x_np, return_layers=True)
    from yolo_amd.engine import Plan
    plan = Plan(model, B, H, keep_all=True)         # no buffer reuse: every layer stays readable
    det_keep = plan.run(torch.from_numpy(x_np).cuda())
    torch.cuda.synchronize()
    assert torch.equal(det_keep, det), "buffer placement changes the result"
    worst = 0.0
    want16, outs16 = oracle_forward(sd, H, C, A, x_np, return_layers=True, emulate_bf16=True)
    for op in plan.ops:
        i = op["index"]
        ref = outs[i + 1] if op["res"] else outs[i]          # fused shortcut: compare after the add
        ref = ref.permute(0, 2, 3, 1).numpy()
        got = plan.layer_output(i).float().cpu().numpy()
        err = _rel_err(got, ref)
        worst = max(worst, err)
        assert err < FP32_ORACLE_LAYER_GATE, f"conv block {i}: relative L2 error {err:.4f} vs the fp32 oracle"
        # ... and no further from the fp32 reference than bf16 storage alone puts the oracle
        ref16 = (outs16[i + 1] if op["res"] else outs16[i]).permute(0, 2, 3, 1).numpy()
        assert err < STORAGE_SLACK * _rel_err(ref16, ref) + 1e-3, f"conv block {i}: {err:.4f} vs {_rel_err(ref16, ref):.4f}"
    _check_teacher_forced(plan, outs16, case["name"])
    got = det.cpu().numpy()
    want = want.numpy()
    if case["name"] in ARR:
        np.testing.assert_allclose(want, ARR[case["name"]], rtol=1e-5, atol=1e-5)   # oracle == reference
    rel = np.abs(got - want) / (np.abs(want) + 1e-3)
    med, p99, mx = float(np.median(rel)), float(np.percentile(rel, 99)), float(rel.max())
    print(f"\n{case['name']}: worst layer rel-L2 {worst:.4f}; output rel err median {med:.2e} "
          f"p99 {p99:.2e} max {mx:.2e}; obj abs err max {np.abs(got[..., 4] - want[..., 4]).max():.2e}")
    assert med < 1e-2 and p99 < 6e-2
    assert np.abs(got[..., 4] - want[..., 4]).max() < 0.05


@pytest.mark.parametrize("conv3", ["0", "1"], ids=["im2col", "patch-reuse"])
def test_forward_416_vs_oracle_and_end_to_end(conv3, monkeypatch):
    """Whole network at full size, with the stride-1 3x3 layers forced through each of
    their two kernels (at batch 32 the large maps take the patch-reuse one by default)."""
    from gpu_helpers import build_model, oracle_forward
    from yolo_amd.utilis import filter_results
    set_knob(monkeypatch, "YOLO_AMD_CONV3", conv3)
    model, params, sd = build_model(416)
    x_np = gi.images(9, 1, 416)
    det = model(torch.from_numpy(x_np).cuda())
    want = oracle_forward(sd, 416, 80, 3, x_np).numpy()
    got = det.cpu().numpy()
    assert got.shape == (1, 10647, 85)
    rel = np.abs(got - want) / (np.abs(want) + 1e-3)
    print(f"\n416: output rel err median {np.median(rel):.2e} p99 {np.percentile(rel, 99):.2e}; "
          f"obj abs err max {np.abs(got[..., 4] - want[..., 4]).max():.2e}")
    assert np.median(rel) < 1e-2
    # NMS exactness is defined at the filter_results input: identical decoded
    # tensor in -> identical rows out
    out = filter_results(det, params)
    assert_bit_equal(out.cpu().numpy(), c_filter_results(got, 0.5, 0.4, 80), "e2e rows given HIP decode")
    # and the bf16-vs-fp32 keep-set agreement is reported, not gated
    ref_rows = c_filter_results(want, 0.5, 0.4, 80)
    print(f"416 e2e: HIP bf16 path keeps {out.shape[0]} rows, fp32 oracle path {ref_rows.shape[0]}")


# ------------------------------------------------------------------ the benchmark configurations
# bf16 bar: boxes and scores within 1e-2 relative of the reference path.  The device path stores
# weights and every layer's output as bf16, and a network of 75 such layers is a chaotic map at
# the level of the last bit: two computations that differ only in the ORDER of their fp32 sums
# part ways (a sum that falls on the other side of a rounding boundary moves an activation by
# 2^-8, which moves the sums of the next layer, ...) until they are as far from each other as
# either is from the fp32 result.  So "within 1e-2 of the reference" is not something a bf16
# path can be held to element by element, and neither is "equal to another bf16 computation".
# What IS testable, and is gated here, is that the kernels add nothing to what bf16 storage costs:
#
# (1) Layer by layer on identical operands ("teacher forcing").  The oracle with bf16 storage
#     (folded bf16 weights, every fused layer's output rounded once to bf16, fp32 sums, fp32
#     detection convolutions and decode) supplies every launch's inputs; the launch's output must
#     then carry the oracle's BITS except where the two fp32 sums straddle a rounding boundary:
#     at most TF_MISMATCH of the elements differ, none by more than one bf16 step, relative L2
#     below TF_REL_L2; fp32 outputs (the detection convolutions) within TF_FP32_REL.
# (2) Free running, end to end: the device path is no further from the fp32 oracle (= the
#     reference) than STORAGE_SLACK x the distance of the oracle with bf16 storage, layer by
#     layer and on the share of output elements inside 1e-2 relative
#     (rel = |got - want| / (|want| + 1e-3)); per layer at most FP32_ORACLE_LAYER_GATE.
# NMS rows are bit-exact given the HIP prediction.  Everything measured -- including how far the
# device path and the oracle with bf16 storage are from each other -- goes to
# gpurun_out/parity_metrics.json (a copy is committed under profiles/).
CONFIG_CASES = [
    # name, H, B, C, A
    ("cfg2_416_b32", 416, 32, 80, 3),         # the benchmark configuration: batch-32 dispatch
    ("cfg3_608_b32", 608, 32, 80, 3),         # one GPU's share of the 8-GPU configuration: 19 x 10 patch tiles on every map
    ("cfg3_608_b2", 608, 2, 80, 3),           # 19 * 2^n maps at a small batch: ragged im2col tiles
    ("cfg5a_320_b1", 320, 1, 80, 3),          # video latency: split reductions
    ("cfg5b_320_b1_a1c4", 320, 1, 4, 1),      # one-anchor variant
]
TF_MISMATCH = 0.01          # measured: at most 0.4 % of a layer's elements (the 4608-long sums), 0.025 % on average
TF_REL_L2 = 5e-4            # measured: at most 1.3e-4
TF_FP32_REL = 1e-3
# |got - ref| <= TF_WORST_STEPS x (2^-7 |ref| + 1e-4 rms(ref)): one bf16 step is between 2^-8 and 2^-7 of
# the value, and near zero the noise of a long fp32 sum (about 1e-6 of its largest partial sum) spans
# several of the then tiny steps (measured: up to 4.0 on the 4608-long sums); a wrong tap, channel or
# border pixel is off by about rms(ref): thousands of these units
TF_WORST_STEPS = 8.0
STORAGE_SLACK = 1.25
FP32_ORACLE_LAYER_GATE = 2e-2


def _check_teacher_forced(keep, outs16, what):
    from gpu_helpers import teacher_forced_layers
    stats = teacher_forced_layers(keep, outs16)
    assert len(stats) == len(keep.ops)
    bad = []
    for st in stats:
        where = f"conv block {st['block']}"
        if not st["rel_l2"] < TF_REL_L2:
            bad.append(f"{where}: relative L2 error {st['rel_l2']:.2e}")
        if "mismatch" in st:
            if not st["mismatch"] < TF_MISMATCH:
                bad.append(f"{where}: {st['mismatch']:.4f} of the elements differ")
            if not st["worst"] <= TF_WORST_STEPS:
                bad.append(f"{where}: an element is {st['worst']:.2f} bf16 steps away")
        elif not st["max_rel"] < TF_FP32_REL:
            bad.append(f"{where}: fp32 output off by {st['max_rel']:.2e}")
    bf = [st for st in stats if "mismatch" in st]
    summary = dict(launches=len(stats), worst_mismatch_share=max(st["mismatch"] for st in bf),
                   mean_mismatch_share=float(np.mean([st["mismatch"] for st in bf])),
                   worst_rel_l2=max(st["rel_l2"] for st in stats),
                   worst_distance_in_bf16_steps=max(st["worst"] for st in bf),
                   worst_fp32_layer_rel=max((st["max_rel"] for st in stats if "max_rel" in st), default=0.0))
    print(f"{what}: every launch on the oracle's bf16 operands: {summary}")
    assert not bad, f"{what}, on the oracle's bf16 operands: " + "; ".join(bad[:12])
    return summary


def _layer_errors(keep, outs, other=None):
    """Relative L2 distance of every conv launch's output (after the fused shortcut add) from
    the oracle's per-block outputs `outs`, computed on the GPU in float64; with `other` (a
    second list of per-block outputs) the distance of THAT from `outs` as well."""
    errs, errs_other = {}, {}
    for op in keep.ops:
        i = op["index"]
        pick = (lambda o: (o[i + 1] if op["res"] else o[i]).permute(0, 2, 3, 1).contiguous().cuda())
        ref = pick(outs)
        den = torch.linalg.vector_norm(ref.flatten(), dtype=torch.float64) + 1e-30
        got = keep.layer_output(i).float()
        errs[i] = float(torch.linalg.vector_norm((got - ref).flatten(), dtype=torch.float64) / den)
        if other is not None:
            errs_other[i] = float(torch.linalg.vector_norm((pick(other) - ref).flatten(), dtype=torch.float64) / den)
        del ref, got
    return errs, errs_other


@pytest.mark.parametrize("case", CONFIG_CASES, ids=lambda c: c[0])
def test_forward_benchmark_configurations(case):
    from gpu_helpers import build_model, keep_set, oracle_forward, record_metric
    from yolo_amd import _lib
    from yolo_amd.engine import Plan
    from yolo_amd.utilis import filter_results
    name, H, B, C, A = case
    model, params, sd = build_model(H, C, A)
    x_np = gi.images(21, B, H)
    x = torch.from_numpy(x_np).cuda()
    det = model(x)
    torch.cuda.synchronize()
    N = A * ((H // 32) ** 2 + (H // 16) ** 2 + (H // 8) ** 2)
    assert tuple(det.shape) == (B, N, 5 + C)

    # which kernels the default plan of this configuration runs
    plan = model._plan(B, H)
    kinds, patch_small_maps = {}, 0
    for j in range(len(plan.ops)):
        ch = _lib.conv_choice(plan.descs[j])
        key = (ch["kernel"], ch["bn"], ch["bk"], ch["act_ksteps"]) if ch["kernel"] == "igemm" else (ch["kernel"],)
        kinds[key] = kinds.get(key, 0) + 1
        if ch["kernel"] == "patch" and plan.ops[j]["Ho"] in (19, 38):
            patch_small_maps += 1
    print(f"\n{name}: kernels {kinds}")
    if name == "cfg2_416_b32":
        assert kinds.get(("igemm", 128, 32, 2), 0) >= 10, "batch 32 was meant to run the 128-channel-tile kernel"
        assert kinds.get(("patch",), 0) >= 25 and kinds.get(("stem",), 0) == 1 and kinds.get(("thin",), 0) == 1
    if name == "cfg3_608_b2":
        # 19 x 10 tiles on the 152^2 and 76^2 maps; the smaller maps have too few tiles at batch 2
        assert kinds.get(("patch",), 0) >= 8
    if name == "cfg3_608_b32":
        assert kinds.get(("patch",), 0) >= 25
        assert patch_small_maps >= 8, "at batch 32 the 19^2 / 38^2 maps were meant to take the 19 x 10 patch tiles"

    keep = Plan(model, B, H, keep_all=True)
    det_keep = keep.run(x)
    torch.cuda.synchronize()
    # the layer-by-layer plan runs the first two layers separately and its own decode
    # launch: same values as the fused plan
    assert torch.equal(det_keep, det), "fused stem / fused decode change the result"
    got = det.cpu().numpy()

    want_t, outs = oracle_forward(sd, H, C, A, x_np, return_layers=True)                        # the reference's arithmetic
    want16_t, outs16 = oracle_forward(sd, H, C, A, x_np, return_layers=True, emulate_bf16=True)  # + bf16 storage
    errs32, errs16_32 = _layer_errors(keep, outs, outs16)      # device path vs fp32; bf16-storage oracle vs fp32
    errs16, _ = _layer_errors(keep, outs16)                    # device path vs bf16-storage oracle (reported)
    del outs
    for i, err in errs32.items():
        assert err < FP32_ORACLE_LAYER_GATE, f"conv block {i}: relative L2 error {err:.4f} vs the fp32 oracle"
        assert err < STORAGE_SLACK * errs16_32[i] + 1e-3, \
            f"conv block {i}: {err:.4f} from the fp32 oracle, bf16 storage alone accounts for {errs16_32[i]:.4f}"
    forced = _check_teacher_forced(keep, outs16, name)          # (overwrites the plan's buffers)
    del outs16, keep
    want, want16 = want_t.numpy(), want16_t.numpy()

    def shares(a, ref):
        rel = np.abs(a - ref) / (np.abs(ref) + 1e-3)
        return rel, dict(all=float((rel <= 1e-2).mean()), boxes=float((rel[..., :4] <= 1e-2).mean()),
                         objectness=float((rel[..., 4] <= 1e-2).mean()), scores=float((rel[..., 5:] <= 1e-2).mean()))
    rel, share32 = shares(got, want)                 # device path vs the reference
    rel_o, share_o = shares(want16, want)            # bf16-storage oracle vs the reference
    rel16, share16 = shares(got, want16)             # device path vs bf16-storage oracle
    obj_err = float(np.abs(got[..., 4] - want[..., 4]).max())
    obj_err_o = float(np.abs(want16[..., 4] - want[..., 4]).max())

    out = filter_results(det, params)
    rows = c_filter_results(got, params["confidence"], params["nms_thesh"], C)
    if rows.shape[0] == 0:
        assert isinstance(out, int) and out == 0
    else:
        assert_bit_equal(out.cpu().numpy(), rows, f"{name}: NMS rows given the HIP prediction")
    ref_rows = c_filter_results(want, params["confidence"], params["nms_thesh"], C)
    ref16_rows = c_filter_results(want16, params["confidence"], params["nms_thesh"], C)
    ks_hip, ks_ref, ks_ref16 = keep_set(got, rows), keep_set(want, ref_rows), keep_set(want16, ref16_rows)

    def jaccard(p, q):
        return round(len(p & q) / max(1, len(p | q)), 5)
    w = want[..., 4]
    summary = lambda r, sh: dict(share_within_1e2=round(sh["all"], 5), share_within_1e2_boxes=round(sh["boxes"], 5),
                                 share_within_1e2_objectness=round(sh["objectness"], 5),
                                 share_within_1e2_scores=round(sh["scores"], 5), median_rel=float(np.median(r)),
                                 p99_rel=float(np.percentile(r, 99)))
    metrics = dict(H=H, B=B, C=C, A=A,
                   layers_on_oracle_operands=forced,
                   device_vs_fp32_oracle=dict(summary(rel, share32), worst_layer_rel_l2=round(max(errs32.values()), 5),
                                              objectness_abs_err_max=obj_err, keep_set_jaccard=jaccard(ks_hip, ks_ref)),
                   bf16_storage_oracle_vs_fp32_oracle=dict(summary(rel_o, share_o),
                                                           worst_layer_rel_l2=round(max(errs16_32.values()), 5),
                                                           objectness_abs_err_max=obj_err_o,
                                                           keep_set_jaccard=jaccard(ks_ref16, ks_ref)),
                   device_vs_bf16_storage_oracle=dict(summary(rel16, share16),
                                                      worst_layer_rel_l2=round(max(errs16.values()), 5),
                                                      first_layers_rel_l2=[round(errs16[op_i], 6) for op_i in sorted(errs16)[:12]],
                                                      keep_set_jaccard=jaccard(ks_hip, ks_ref16)),
                   objectness_fp32_oracle=dict(min=float(w.min()), p01=float(np.percentile(w, 1)),
                                               median=float(np.median(w)), p99=float(np.percentile(w, 99)),
                                               max=float(w.max()),
                                               share_within_0p02_of_conf=float(
                                                   (np.abs(w - params["confidence"]) <= 0.02).mean())),
                   rows=dict(device=int(rows.shape[0]), fp32_oracle=int(ref_rows.shape[0]),
                             bf16_storage_oracle=int(ref16_rows.shape[0])),
                   kernels={"/".join(map(str, k)): v for k, v in kinds.items()})
    print(f"{name}: {metrics}")
    record_metric(name, metrics)
    # end to end, the device path is as close to the reference as bf16 storage allows
    assert share32["all"] >= share_o["all"] - 0.03, \
        f"{share32['all']:.3f} of the output within 1e-2 of the reference; bf16 storage alone gives {share_o['all']:.3f}"
    assert np.median(rel) < max(1e-2, STORAGE_SLACK * float(np.median(rel_o))) and obj_err < max(0.05, 2 * obj_err_o)
    assert np.median(rel) < 1e-2 and obj_err < 0.05


def test_batch_consistency_and_plan_reuse(monkeypatch):
    from gpu_helpers import build_model
    model, params, _ = build_model(96)
    x = torch.from_numpy(gi.images(11, 4, 96)).cuda()
    full = model(x)
    again = model(x)
    assert torch.equal(full, again), "the forward pass is not deterministic"
    # At such a small batch the deep layers split their reductions, into a number of
    # slices that depends on the batch: the last bits may then differ between batch
    # sizes, nothing more.
    halves = torch.cat((model(x[:2]), model(x[2:])), 0)
    rel = ((full - halves).abs() / (full.abs() + 1e-3)).flatten()
    assert rel.median() < 2e-3 and rel.quantile(0.999) < 0.1
    # With every reduction kept in one pass, per-image results are bit-identical
    # whatever batch they are in.
    set_knob(monkeypatch, "YOLO_AMD_SPLITK", "0")
    model.refold()                                  # new plans
    full = model(x)
    halves = torch.cat((model(x[:2]), model(x[2:])), 0)
    assert torch.equal(full, halves), "per-image results depend on the batch they are in"


SPLITK_CASES = [
    # name, B, H, c0, cout, k, stride, leaky, residual, fp32_out   (split needs >= 64 K-steps of 32)
    ("k3_c512_n1024_13_res", 1, 13, 512, 1024, 3, 1, True, True, False),     # 144 K-steps, 32 tiles
    ("k3_c256_n512_26_res", 1, 26, 256, 512, 3, 1, True, True, False),
    ("k1_c2048_n512_13", 1, 13, 2048, 512, 1, 1, True, False, False),
    ("k3s2_c256_n512_26", 2, 26, 256, 512, 3, 2, True, False, False),
    ("head_c2048_n255_13", 1, 13, 2048, 255, 1, 1, False, False, True),      # fp32 out, odd channel count
    ("k1_c2144_n128_ragged", 1, 15, 2144, 128, 1, 1, True, False, False),    # 67 K-steps: a short last slice, BK = 32
    ("k3_c352_n64_bk32", 1, 13, 352, 64, 3, 1, True, True, False),           # 99 K-steps of 32, one-step ring
]


@pytest.mark.parametrize("case", SPLITK_CASES, ids=lambda c: c[0])
def test_split_reduction_matches_the_reference_and_the_single_pass(case):
    """Layers with few output tiles and a long K, with the scratch lent: the slices'
    partial sums plus the finishing kernel must give what the fp32 reference gives,
    to the single-pass tolerance, and stay within rounding of the single pass."""
    from gpu_helpers import conv_reference, run_conv
    name, B, H, c0, cout, k, stride, leaky, use_res, fp32_out = case
    r = gi.rs(case_seed(name))
    x0 = torch.from_numpy(r.standard_normal((B, H, H, c0)).astype(np.float32)).cuda().to(torch.bfloat16)
    w = torch.from_numpy((r.standard_normal((cout, c0, k, k)) * np.sqrt(2.0 / (c0 * k * k))).astype(np.float32))
    bias = torch.from_numpy(r.normal(0, 0.3, cout).astype(np.float32))
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    res = None
    if use_res:
        res = torch.from_numpy(r.standard_normal((B, Ho, Ho, cout)).astype(np.float32)).cuda().to(torch.bfloat16)
    info = {}
    split = run_conv(x0, w, bias, stride=stride, leaky=leaky, residual=res, fp32_out=fp32_out, splitk=True,
                     info=info, repeat=3).float().cpu()
    assert info["splitk_bytes"] > 0, "this shape was meant to be split"
    assert info["choice"]["k_slices"] > 1
    single = run_conv(x0, w, bias, stride=stride, leaky=leaky, residual=res, fp32_out=fp32_out).float().cpu()
    want = conv_reference(x0, w, bias, stride=stride, leaky=leaky, residual=res)
    assert torch.isfinite(split).all(), "output holds NaN/inf (unwritten elements or scratch read before written)"
    tol = 2e-4 if fp32_out else 1.0 / 128
    torch.testing.assert_close(split, want, rtol=tol, atol=tol)
    torch.testing.assert_close(split, single, rtol=tol, atol=tol)


def test_large_batches_are_not_split():
    from yolo_amd._lib import ConvDesc, lib
    d = (ConvDesc * 1)()
    d[0].c0, d[0].c1, d[0].B, d[0].H, d[0].W = 512, 0, 32, 13, 13
    d[0].cout, d[0].cout_pad, d[0].ksize, d[0].stride, d[0].k_pad = 1024, 1024, 3, 1, 4608
    assert lib.yolo_amd_conv_splitk_bytes(d) == 0
    d[0].B = 1
    assert lib.yolo_amd_conv_splitk_bytes(d) >= 2 * 169 * 1024 * 4


def test_split_heads_decode_like_the_single_pass():
    """Batch 1 at 416: the detection convolutions are split and the finishing kernel
    decodes; the prediction must equal the unsplit plan's to fp32 rounding."""
    from gpu_helpers import build_model
    model, params, _ = build_model(416, C=80, seed=5)
    x = torch.from_numpy(gi.images(13, 1, 416)).cuda()
    with torch.no_grad():
        split = model(x).clone()
    assert model._plan(1, 416).splitk_bytes > 0
    import os
    os.environ["YOLO_AMD_SPLITK"] = "0"
    try:
        from yolo_amd._lib import lib
        lib.yolo_amd_reload_knobs()
        model.refold()
        with torch.no_grad():
            single = model(x)
        assert model._plan(1, 416).splitk_bytes == 0
    finally:
        del os.environ["YOLO_AMD_SPLITK"]
        lib.yolo_amd_reload_knobs()
    # 75 layers of bf16 activations: a different summation order flips last bits
    # (2^-8 relative each) here and there, and those differences travel on
    rel = ((split - single).abs() / (single.abs() + 1e-3)).flatten()
    assert rel.median() < 1e-2 and (split[..., 4] - single[..., 4]).abs().max() < 0.05
