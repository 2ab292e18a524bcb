"""Pin the CPU oracle (oracle/yolo_oracle.py and oracle/nms_oracle.c) against
the golden vectors captured from the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import gen_inputs as gi
import yolo_oracle as orc
from helpers import (assert_bit_equal, c_box_iou, c_decode_head, c_filter_results,
                     load_golden, nms_case_input, sha)
from yolo_amd.cfg import default_params, yolov3_cfg_text

META, ARR = load_golden()
NMS_CASES = {c["name"]: c for c in META["nms_cases"]}
SMALL = [n for n, c in NMS_CASES.items() if not n.startswith("full")]
FULL = [n for n in NMS_CASES if n.startswith("full")]


def test_cfg_layer_table_matches_reference_modules():
    blocks = orc.parse_cfg_text(yolov3_cfg_text())
    assert len(blocks) == META["n_blocks"] == 107
    table = orc.layer_table(blocks, 3, 80, 3)
    convs = [[i, e["cin"], e["cout"], e["k"], e["stride"], e["pad"], not e["bn"],
              1 + int(e["bn"]) + int(e["leaky"])]
             for i, e in enumerate(table) if e["type"] == "conv"]
    assert convs == META["conv_table"]
    ltd = META["layer_type_dic"]
    assert [i for i, e in enumerate(table) if e["type"] == "conv"] == ltd["conv"]
    assert [i for i, e in enumerate(table) if e["type"] == "shortcut"] == ltd["shortcut"]
    assert [i for i, e in enumerate(table) if e["type"] == "yolo"] == ltd["yolo"] == [82, 94, 106]
    assert [i for i, e in enumerate(table) if e["type"] == "route" and len(e["srcs"]) == 1] == ltd["route_1"] == [83, 95]
    assert [i for i, e in enumerate(table) if e["type"] == "route" and len(e["srcs"]) == 2] == ltd["route_2"] == [86, 98]
    rel = ltd["referred_relationship"]
    assert rel["83"] == 79 and rel["86"] == [85, 61] and rel["95"] == 91 and rel["98"] == [97, 36]
    n_params = sum(e["cout"] * e["cin"] * e["k"] ** 2 + (2 * e["cout"] if e["bn"] else e["cout"])
                   for e in table if e["type"] == "conv")
    assert n_params == META["num_parameters"] == 61949149


@pytest.mark.parametrize("case", META["decode_cases"], ids=lambda c: c["name"])
def test_decode_oracles_match_reference(case):
    H, C, A, B = case["H"], case["C"], case["A"], case["B"]
    p = default_params(H, C, A)
    py_out, c_out = [], []
    for k, s in enumerate((32, 16, 8)):
        raw = gi.head_raw(100 + 10 * k + H, B, A, C, H // s)
        anc = p["anchors"][k * A:(k + 1) * A]
        py_out.append(orc.decode_head(torch.from_numpy(raw), anc, H, C).numpy())
        c_out.append(c_decode_head(raw, anc, H, C))
    want = ARR[case["name"]]
    # 1e-5 relative: torch's own sigmoid differs by 1 ulp between strided and
    # contiguous slices, so bit equality is not available even within torch.
    np.testing.assert_allclose(np.concatenate(py_out, 1), want, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(np.concatenate(c_out, 1), want, rtol=1e-5, atol=1e-6)


def test_decode_overflow_matches_reference():
    p = default_params(64, 80, 3)
    raw = gi.head_raw(7, 1, 3, 80, 2)
    raw[0, 2, 0, 0] = 100.0
    raw[0, 85 + 3, 1, 1] = 95.0
    want = ARR["dec_overflow"]
    assert np.isinf(want).sum() == 2
    for got in (orc.decode_head(torch.from_numpy(raw), p["anchors"][0:3], 64, 80).numpy(),
                c_decode_head(raw, p["anchors"][0:3], 64, 80)):
        assert np.array_equal(np.isinf(got), np.isinf(want))
        fin = np.isfinite(want)
        np.testing.assert_allclose(got[fin], want[fin], rtol=1e-5, atol=1e-6)


def test_box_iou_bit_exact():
    a, b = gi.iou_pairs(11)
    assert_bit_equal(orc.box_iou(a, b), ARR["iou_pairs"], "numpy oracle pairs")
    assert_bit_equal(c_box_iou(a, b), ARR["iou_pairs"], "C oracle pairs")
    assert_bit_equal(orc.box_iou(a[100:101], b), ARR["iou_bcast"], "numpy oracle broadcast")
    assert_bit_equal(c_box_iou(a[100:101], b), ARR["iou_bcast"], "C oracle broadcast")
    assert np.isnan(ARR["iou_pairs"][:12]).any()          # the special cases bite


@pytest.mark.parametrize("name", SMALL)
def test_filter_results_small_bit_exact(name):
    case = NMS_CASES[name]
    pred = nms_case_input(case)
    C = case["gen"]["C"]
    py = orc.filter_results(pred, case["conf"], case["nms"], C)
    c = c_filter_results(pred, case["conf"], case["nms"], C)
    if case["K"] == 0:
        assert isinstance(py, int) and py == 0 and case["sentinel_int"] == 0
        assert c.shape[0] == 0
        return
    want = ARR["nms_" + name]
    assert want.shape == (case["K"], 9)
    assert_bit_equal(py, want, "numpy oracle")
    assert_bit_equal(c, want, "C oracle")
    assert np.array_equal(want[:, 8], want[:, 2]) or np.isnan(want[:, 8]).any()


@pytest.mark.parametrize("name", FULL)
def test_filter_results_full_size_hash(name):
    case = NMS_CASES[name]
    pred = nms_case_input(case)
    c = c_filter_results(pred, case["conf"], case["nms"], case["gen"]["C"])
    assert c.shape[0] == case["K"]
    assert sha(c) == case["sha256"]
    np.testing.assert_array_equal(c[:3], np.array(case["first_rows"], np.float32))


def test_torch_loop_port_equals_oracle():
    case = NMS_CASES["clustered_n600_c4_conf30"]
    pred = nms_case_input(case)
    out = orc.filter_results_torch_loop(torch.from_numpy(pred), case["conf"], case["nms"], 4)
    assert_bit_equal(out.numpy(), ARR["nms_" + case["name"]], "torch-loop port")
    empty = orc.filter_results_torch_loop(torch.from_numpy(nms_case_input(NMS_CASES["all_below"])),
                                          0.5, 0.4, 80)
    assert isinstance(empty, int) and empty == 0


@pytest.mark.parametrize("case", META["forward_cases"], ids=lambda c: c["name"])
def test_forward_oracle_matches_reference(case):
    H, B, C, A = case["H"], case["B"], case["C"], case["A"]
    blocks = orc.parse_cfg_text(yolov3_cfg_text(C, A))
    p = default_params(H, C, A)
    table = orc.layer_table(blocks, 3, C, A)
    sd = {k: torch.from_numpy(np.asarray(v))
          for k, v in gi.synthetic_state_dict(table, case["weight_seed"]).items()}
    x = torch.from_numpy(gi.images(case["image_seed"], B, H))
    det, outs = orc.forward(sd, blocks, p, x, return_layers=True)
    det = det.numpy()
    assert list(det.shape) == case["out_shape"]
    if "fwd_" + "h64" in case["name"]:
        np.testing.assert_allclose(det, ARR[case["name"]], rtol=1e-5, atol=1e-5)
    else:
        idx = np.array(case["out_samples_idx"])
        np.testing.assert_allclose(det.reshape(-1)[idx], np.array(case["out_samples"], np.float32),
                                   rtol=1e-5, atol=1e-5)
    # conv outputs BEFORE the in-place shortcut add (the reference's hook sees them there)
    for i, dg in case["conv_digests"].items():
        i = int(i)
        if table[i + 1]["type"] == "shortcut" if i + 1 < len(table) else False:
            continue        # oracle keeps only the post-add tensor for these
        o = outs[i]
        assert list(o.shape) == dg["shape"]
        np.testing.assert_allclose(float(o.double().abs().sum()), dg["abssum"], rtol=1e-5)
    # end to end through NMS, exact rows given the oracle's own decode output
    got = orc.filter_results(ARR[case["name"]] if case["name"] in ARR else det, 0.5, 0.4, C)
    if case["name"] in ARR:
        assert_bit_equal(got, ARR[case["name"] + "_e2e"], "e2e rows")


def test_forward_oracle_with_bf16_storage():
    """emulate_bf16 changes nothing but the storage roundings: every stored activation is a
    bf16 value, every layer stays within bf16 distance of the fp32 oracle, and the default
    (fp32, pinned to the reference above) is untouched by the option."""
    case = next(c for c in META["forward_cases"] if "h64" in c["name"])
    H, B, C, A = case["H"], case["B"], case["C"], case["A"]
    blocks = orc.parse_cfg_text(yolov3_cfg_text(C, A))
    p = default_params(H, C, A)
    table = orc.layer_table(blocks, 3, C, A)
    sd = {k: torch.from_numpy(np.asarray(v))
          for k, v in gi.synthetic_state_dict(table, case["weight_seed"]).items()}
    x = torch.from_numpy(gi.images(case["image_seed"], B, H))
    det32, outs32 = orc.forward(sd, blocks, p, x, return_layers=True)
    det16, outs16 = orc.forward(sd, blocks, p, x, return_layers=True, emulate_bf16=True)
    np.testing.assert_allclose(det32.numpy(), ARR[case["name"]], rtol=1e-5, atol=1e-5)
    assert det16.shape == det32.shape and not torch.equal(det16, det32)
    n_checked = 0
    for i, ent in enumerate(table):
        if ent["type"] not in ("conv", "shortcut") or outs16[i] is None:
            continue
        nxt = table[i + 1]["type"] if i + 1 < len(table) else None
        o16, o32 = outs16[i], outs32[i]
        stored = ent["type"] == "shortcut" or nxt not in ("shortcut", "yolo")
        if stored:       # what the device path writes to memory: exactly representable in bf16
            assert torch.equal(o16.to(torch.bfloat16).float(), o16), f"block {i} holds non-bf16 values"
            n_checked += 1
        err = float((o16 - o32).double().norm() / o32.double().norm())
        assert err < 2e-2, f"block {i}: bf16 storage moved the layer by {err:.4f} relative L2"
    assert n_checked >= 72
    rel = (det16 - det32).abs() / (det32.abs() + 1e-3)
    assert float(rel.median()) < 1e-2
    with pytest.raises(ValueError):
        orc.forward(sd, blocks, p, x, training=True, emulate_bf16=True)


def test_two_bf16_computations_that_differ_only_in_their_sums_part_ways():
    """Why a bf16 path cannot be held to another bf16 computation element by element.  The
    oracle with bf16 storage, once with float32 and once with float64 convolution sums: the
    same network, the same roundings, sums that differ in their last bit (1e-7 relative).  A
    sum that falls on the other side of a bf16 rounding boundary moves an activation by 2^-8,
    which moves the next layer's sums, and so on: within a few dozen layers the two are about
    as far from each other as either is from the fp32 network, while both stay equally close
    to THAT.  (The GPU tests gate the device path the same way: against the fp32 oracle,
    relative to what bf16 storage alone costs, and bit by bit only layer by layer on
    identical operands.)"""
    H, B, C, A = 96, 1, 80, 3
    blocks = orc.parse_cfg_text(yolov3_cfg_text(C, A))
    p = default_params(H, C, A)
    table = orc.layer_table(blocks, 3, C, A)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in gi.synthetic_state_dict(table, 3).items()}
    x = torch.from_numpy(gi.images(21, B, H))
    _, o32 = orc.forward(sd, blocks, p, x, return_layers=True)
    _, oa = orc.forward(sd, blocks, p, x, return_layers=True, emulate_bf16=True)
    _, ob = orc.forward(sd, blocks, p, x, return_layers=True, emulate_bf16=True, sum_dtype=torch.float64)
    convs = [i for i, e in enumerate(table) if e["type"] == "conv" and table[i + 1]["type"] != "yolo"]
    rel = lambda u, v: float((u - v).double().norm() / v.double().norm())
    ab = [rel(oa[i], ob[i]) for i in convs]
    a32 = [rel(oa[i], o32[i]) for i in convs]
    b32 = [rel(ob[i], o32[i]) for i in convs]
    assert ab[0] < 1e-4                                   # they start out (nearly) identical ...
    assert max(ab[40:]) > 2e-3                            # ... and end up a rounding step apart on average
    assert max(ab) < 1.5 * max(a32)                       # never further apart than from the fp32 network
    assert abs(max(a32) - max(b32)) < 0.3 * max(a32)      # to which both are equally close
    assert max(a32) < 2e-2


# ---- train() mode: BatchNorm2d on batch statistics (video.py / get_fps.py never call eval())
import json as _json
import os as _os

_TRAIN_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden")
TRAIN_META = _json.load(open(_os.path.join(_TRAIN_DIR, "golden_train.json")))
TRAIN_ARR = np.load(_os.path.join(_TRAIN_DIR, "golden_train.npz"))


@pytest.mark.parametrize("case", TRAIN_META["cases"], ids=lambda c: c["name"])
def test_train_mode_forward_oracle_matches_reference(case):
    """Two calls in train() mode (outputs, moved running statistics), then the eval()
    model on the moved statistics, against what the reference returned
    (tests/golden/make_golden_train.py)."""
    H, B, C, A = case["H"], case["B"], case["C"], case["A"]
    blocks = orc.parse_cfg_text(yolov3_cfg_text(C, A))
    p = default_params(H, C, A)
    table = orc.layer_table(blocks, 3, C, A)
    sd = {k: torch.from_numpy(np.asarray(v))
          for k, v in gi.synthetic_state_dict(table, TRAIN_META["weight_seed"]).items()}
    before = {k: v.clone() for k, v in sd.items()}
    for k, seed in enumerate(TRAIN_META["image_seeds"]):
        det = orc.forward(sd, blocks, p, torch.from_numpy(gi.images(seed, B, H)), training=True).numpy()
        np.testing.assert_allclose(det, TRAIN_ARR[f"{case['name']}_call{k}"], rtol=2e-4, atol=2e-4)
    n_checked = 0
    for key, dg in case["bn"].items():
        a = sd[key].double().numpy().reshape(-1)
        assert not np.array_equal(a, before[key].double().numpy().reshape(-1)), f"{key} did not move"
        np.testing.assert_allclose(np.abs(a).sum(), dg["abssum"], rtol=1e-5)
        idx = np.linspace(0, a.size - 1, min(4, a.size)).astype(np.int64)
        np.testing.assert_allclose(a[idx], dg["samples"], rtol=1e-4, atol=1e-6)
        n_checked += 1
    assert n_checked == 2 * 72
    nbt = [int(v) for k, v in sd.items() if k.endswith("num_batches_tracked")]
    assert nbt and all(n == case["num_batches_tracked"] for n in nbt)
    det = orc.forward(sd, blocks, p, torch.from_numpy(gi.images(TRAIN_META["image_seeds"][0], B, H))).numpy()
    np.testing.assert_allclose(det, TRAIN_ARR[f"{case['name']}_eval_after"], rtol=2e-4, atol=2e-4)
