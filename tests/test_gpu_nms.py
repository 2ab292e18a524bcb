"""HIP filter_results / filter_img_class / box_iou against the golden vectors
(captured from the reference) and the C oracle.  Bit-exact: integer/index work
and fp32 IoU arithmetic in the reference's order leave no tolerance."""
import numpy as np
import pytest
import torch

import gen_inputs as gi
import yolo_oracle as orc
from helpers import (assert_bit_equal, c_filter_results, load_golden, nms_case_input, sha)

pytestmark = pytest.mark.gpu

META, ARR = load_golden()
NMS_CASES = {c["name"]: c for c in META["nms_cases"]}


def hip_filter(pred_np, conf, nms, C, device="cuda"):
    from yolo_amd.utilis import filter_results
    t = torch.from_numpy(pred_np).to(device)
    before = t.clone()
    out = filter_results(t, {"confidence": conf, "nms_thesh": nms, "num_classes": C})
    assert torch.equal(t.view(torch.int32), before.view(torch.int32)), "prediction was modified"
    return out


def test_library_is_the_hip_one():
    from yolo_amd import _lib
    assert torch.cuda.is_available()
    import os, re
    from helpers import ROOT
    header = open(os.path.join(ROOT, "include", "yolo_amd.h")).read()
    assert _lib.lib.yolo_amd_abi_version() == int(re.search(r"#define YOLO_AMD_ABI_VERSION\s+(\d+)", header).group(1))


@pytest.mark.parametrize("name", [n for n in NMS_CASES if not n.startswith("full")])
def test_filter_results_golden(name):
    case = NMS_CASES[name]
    out = hip_filter(nms_case_input(case), case["conf"], case["nms"], case["gen"]["C"])
    if case["K"] == 0:
        assert isinstance(out, int) and out == 0
        return
    assert out.is_cuda and out.dtype == torch.float32
    assert_bit_equal(out.cpu().numpy(), ARR["nms_" + name], name)


@pytest.mark.parametrize("name", [n for n in NMS_CASES if n.startswith("full")])
def test_filter_results_full_size(name):
    case = NMS_CASES[name]
    pred = nms_case_input(case)
    out = hip_filter(pred, case["conf"], case["nms"], case["gen"]["C"]).cpu().numpy()
    assert out.shape[0] == case["K"]
    assert sha(out) == case["sha256"]
    assert_bit_equal(out, c_filter_results(pred, case["conf"], case["nms"], case["gen"]["C"]), name)


def test_more_candidates_than_the_lds_sort_holds():
    # 22743 rows, nearly all pass: the sort runs in global memory
    pred = gi.nms_pred(seed=61, B=2, N=22743, C=80, H=608)
    out = hip_filter(pred, 0.01, 0.4, 80).cpu().numpy()
    assert_bit_equal(out, c_filter_results(pred, 0.01, 0.4, 80), "global sort path")


def test_nms_stress_batch():
    # mAP-eval setting: conf 0.01, 10647 boxes x 80 classes per image
    pred = gi.nms_pred(seed=62, B=8, N=10647, C=80)
    out = hip_filter(pred, 0.01, 0.4, 80).cpu().numpy()
    want = c_filter_results(pred, 0.01, 0.4, 80)
    assert_bit_equal(out, want, "stress batch")
    # size-independent properties: ordering and idempotence of the keep set
    img, cls, obj = out[:, 0], out[:, 7], out[:, 5]
    key = img * 1000 + cls
    assert (np.diff(key) >= 0).all()
    same = np.diff(key) == 0
    assert (np.diff(obj)[same] <= 0).all()


def test_nms_stress_batch_64():
    """The NMS-stress configuration at its full size: batch 64, 10647 boxes x 80
    classes per image, conf 0.01 (474 MB of workspace).  Bit-exact against the C
    oracle on the whole batch, plus the size-independent properties: ordering, every
    row above the threshold, no two kept boxes of one (image, class) with IoU >= the
    threshold, and idempotence (NMS of the kept rows keeps them all)."""
    from helpers import c_box_iou
    from yolo_amd._lib import lib
    B, N, C = 64, 10647, 80
    assert lib.yolo_amd_filter_results_workspace_bytes(B, N, C) > 400 << 20
    pred = gi.nms_pred(seed=68, B=B, N=N, C=C)
    pred[:, ::2] = gi.nms_pred(seed=69, B=B, N=(N + 1) // 2, C=C, mode="clustered")   # half the boxes crowd
    out = hip_filter(pred, 0.01, 0.4, C).cpu().numpy()
    want = c_filter_results(pred, 0.01, 0.4, C)
    assert_bit_equal(out, want, "stress batch 64")
    img, cls, obj = out[:, 0], out[:, 7], out[:, 5]
    key = img * 1000 + cls
    assert (np.diff(key) >= 0).all()
    assert (np.diff(obj)[np.diff(key) == 0] <= 0).all()
    assert (obj > np.float32(0.01)).all() and set(np.unique(img).astype(int)) == set(range(B))
    # pairwise IoU inside a few crowded (image, class) groups
    checked = 0
    for b, c in [(0, 0), (17, 7), (63, 79), (40, 35)]:
        grp = out[(img == b) & (cls == c)][:, 1:5]
        for i in range(min(len(grp) - 1, 40)):
            assert (c_box_iou(grp[i:i + 1], grp[i + 1:]) < np.float32(0.4)).all()
            checked += 1
    assert checked > 20
    # idempotence on the first four images: feed the kept boxes back in
    sub = out[img < 4]
    again = np.zeros((4, int(np.bincount(sub[:, 0].astype(int)).max()), 5 + C), dtype=np.float32)
    for b in range(4):
        rb = sub[sub[:, 0] == b]
        n = len(rb)
        again[b, :n, 0] = (rb[:, 1] + rb[:, 3]) / 2
        again[b, :n, 1] = (rb[:, 2] + rb[:, 4]) / 2
        again[b, :n, 2] = rb[:, 3] - rb[:, 1]
        again[b, :n, 3] = rb[:, 4] - rb[:, 2]
        again[b, :n, 4] = rb[:, 5]
        again[b, np.arange(n), 5 + rb[:, 7].astype(int)] = rb[:, 6]
    out2 = hip_filter(again, 0.01, 0.4, C).cpu().numpy()
    assert_bit_equal(out2, c_filter_results(again, 0.01, 0.4, C), "second pass")
    # (re-centred boxes differ in the last bit, so the second pass may drop a handful)
    assert out2.shape[0] >= 0.99 * sub.shape[0]


def test_two_streams_do_not_share_a_workspace():
    """filter_results keeps one workspace per (device, stream): two calls queued on
    different streams may overlap on the GPU and must still both be right."""
    from yolo_amd.utilis import filter_results_async
    preds = [gi.nms_pred(seed=90 + i, B=4, N=3000, C=80, mode="clustered") for i in range(2)]
    want = [c_filter_results(p, 0.05, 0.4, 80) for p in preds]
    ts = [torch.from_numpy(p).cuda() for p in preds]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    params = {"confidence": 0.05, "nms_thesh": 0.4, "num_classes": 80}
    torch.cuda.synchronize()
    for rep in range(3):
        pending = []
        for t, st in zip(ts, streams):
            with torch.cuda.stream(st):
                pending.append(filter_results_async(t, params))      # nothing waits: the two overlap
        for o, w in zip(pending, want):
            assert_bit_equal(o.result().cpu().numpy(), w, f"stream result, repetition {rep}")
    assert filter_results_async(ts[0][:, :0], params).result() == 0


def test_single_class_large_segment():
    # one class, thousands of boxes in one segment, ties included
    pred = gi.nms_pred(seed=63, B=2, N=5000, C=1, mode="clustered", tie_levels=256)
    out = hip_filter(pred, 0.05, 0.4, 1).cpu().numpy()
    assert_bit_equal(out, c_filter_results(pred, 0.05, 0.4, 1), "single class")
    pred = gi.nms_pred(seed=64, B=1, N=3000, C=1)
    out = hip_filter(pred, 0.3, 0.4, 1).cpu().numpy()      # few overlaps: long kept lists
    assert_bit_equal(out, c_filter_results(pred, 0.3, 0.4, 1), "single class sparse")


def test_unaligned_view_and_per_image_calls():
    # evaluate.py feeds filter_results one image at a time as img.unsqueeze(0):
    # views that start at 4-byte, not 16-byte, aligned addresses
    from yolo_amd.utilis import filter_results
    pred = gi.nms_pred(seed=65, B=3, N=1001, C=7)
    t = torch.from_numpy(pred).cuda()
    params = {"confidence": 0.3, "nms_thesh": 0.4, "num_classes": 7}
    whole = c_filter_results(pred, 0.3, 0.4, 7)
    for b in range(3):
        view = t[b].unsqueeze(0)
        got = filter_results(view, params).cpu().numpy()
        want = whole[whole[:, 0] == b].copy()
        want[:, 0] = 0
        assert_bit_equal(got, want, f"image {b}")


def test_confidence_sweep_like_get_map():
    from yolo_amd.utilis import filter_results
    pred = gi.nms_pred(seed=66, B=1, N=2000, C=4, mode="clustered")
    t = torch.from_numpy(pred).cuda()
    params = {"nms_thesh": 0.4, "num_classes": 4}
    for conf in list(np.arange(start=0.2, stop=0.95, step=0.025))[::4]:
        params["confidence"] = conf
        got = filter_results(t, params)
        want = c_filter_results(pred, float(np.float32(conf)), 0.4, 4)
        if want.shape[0] == 0:
            assert isinstance(got, int) and got == 0
        else:
            assert_bit_equal(got.cpu().numpy(), want, f"conf {conf}")


def test_cpu_tensor_round_trip():
    case = NMS_CASES["clustered_n600_c4_conf30"]
    out = hip_filter(nms_case_input(case), case["conf"], case["nms"], 4, device="cpu")
    assert not out.is_cuda
    assert_bit_equal(out.numpy(), ARR["nms_" + case["name"]], "cpu round trip")


def test_box_iou_golden():
    from yolo_amd.utilis import box_iou
    a, b = gi.iou_pairs(11)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    assert_bit_equal(box_iou(ta, tb).cpu().numpy(), ARR["iou_pairs"], "pairs")
    assert_bit_equal(box_iou(ta[100:101], tb).cpu().numpy(), ARR["iou_bcast"], "broadcast")
    assert box_iou(ta[:1], tb[:0]).shape == (0,)


def test_filter_img_class_matches_oracle():
    from yolo_amd.utilis import filter_img_class
    pred = gi.nms_pred(seed=67, B=1, N=1500, C=5, mode="clustered", tie_levels=128)
    rows, n_index = orc.candidates(pred[0], 0.2, 5)
    t = torch.from_numpy(rows).cuda()
    for cls in np.unique(rows[:, 6]):
        sel = np.nonzero(rows[:, 6] == cls)[0]
        grp = rows[sel]
        order = np.lexsort((sel, -grp[:, 4].astype(np.float64)))
        grp = grp[order]
        want = grp[orc.greedy_nms_sorted(grp, 0.4)]
        ind, got = filter_img_class(float(cls), t, 3, 0.4)
        assert_bit_equal(got.cpu().numpy(), want, f"class {cls}")
        assert ind.shape == (want.shape[0], 1) and (ind == 3).all()
    ind, got = filter_img_class(99.0, t, 0, 0.4)
    assert got.shape == (0, 8) and ind.shape == (0, 1)


def test_thresholds_on_the_rounding_boundary():
    """The suppression test `iou < thresh` is decided without the division (an exact
    float64 multiply-compare against the midpoint below thresh).  Thresholds that EQUAL
    IoUs occurring in the data, their fp32 neighbours, powers of two, and values for which
    the library keeps the division (zero, negative, denormal, inf, NaN); integer boxes so
    that few distinct IoU values occur and many pairs sit exactly on the threshold; some
    boxes with negative width or height (non-positive union)."""
    from helpers import c_box_iou
    C = 3
    pred = gi.nms_pred(seed=71, B=2, N=1500, C=C, mode="clustered", tie_levels=16)
    pred[..., :4] = np.round(pred[..., :4] / 4) * 4
    pred[0, :60, 2] *= -3                      # negative widths: areas and unions <= 0
    pred[1, 100:130, 3] = -1.0                 # height -1: (h + 1) = 0, zero areas
    # IoU values between kept boxes and the other candidates of their class for which a
    # pair sitting EXACTLY on the threshold decides a box's fate
    out = c_filter_results(pred, 0.3, 0.5, C)
    kept0 = out[(out[:, 0] == 1) & (out[:, 7] == 0)][:, 1:5].astype(np.float32)
    rows, _ = orc.candidates(pred[1], 0.3, C)
    cand = rows[rows[:, 6] == 0][:, :4].astype(np.float32)
    vals = np.unique(np.concatenate([c_box_iou(k[None, :], cand) for k in kept0[:20]]))
    vals = vals[(vals > 0.3) & (vals < 0.7)]
    picked = []
    for t in vals:
        up = np.nextafter(np.float32(t), np.float32(1))
        if c_filter_results(pred, 0.3, float(t), C).shape[0] != c_filter_results(pred, 0.3, float(up), C).shape[0]:
            picked.append(t)
        if len(picked) == 3:
            break
    assert len(picked) == 3
    thresholds = []
    for t in picked:
        t = np.float32(t)
        thresholds += [t, np.nextafter(t, np.float32(0)), np.nextafter(t, np.float32(1))]
    thresholds += [np.float32(0.5), np.float32(0.25), np.float32(1.0), np.nextafter(np.float32(1), np.float32(2)),
                   np.float32(1e-3), np.float32(0.0), np.float32(-0.5), np.float32(1e-40), np.float32(np.inf),
                   np.float32(np.nan)]
    kept = set()
    for t in thresholds:
        want = c_filter_results(pred, 0.3, float(t), C)
        got = hip_filter(pred, 0.3, float(t), C)
        if want.shape[0] == 0:
            assert isinstance(got, int) and got == 0
        else:
            assert_bit_equal(got.cpu().numpy(), want, f"nms threshold {t!r}")
        kept.add(want.shape[0])
    assert len(kept) >= 6, "the thresholds did not change the result: the case tests nothing"
