"""CPU checks of the scoring code (yolo_amd.evaluate) against what the reference's
evaluate.py returned on the same seeded data sets (golden/golden_eval.json).  The
detections come from the oracle's filter_results here; the GPU suite repeats the
get_map cases with the HIP path producing them."""
import contextlib
import io

import numpy as np
import pytest
import torch

import gen_inputs as gi
import yolo_oracle as orc
from eval_data import CASES, GOLDEN, ReplayModel, batches, check_against_golden
from yolo_amd import evaluate as ev
from yolo_amd.cfg import default_params


def test_bbox_iou_numpy_matches_the_reference_matrix():
    a, b = gi.iou_pairs(64, m=64)
    with np.errstate(invalid="ignore"):
        got = ev.bbox_iou_numpy(a.astype(np.float64), b.astype(np.float64))
    want = np.array(GOLDEN["bbox_iou_numpy"]["matrix"], dtype=np.float64)
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.int64), want.view(np.int64)) or \
        np.array_equal(got, want, equal_nan=True)


def oracle_filter_multi(prediction, params, conf_list):
    pred = prediction.detach().cpu().numpy()
    lo = min(float(c) for c in conf_list)
    base = orc.filter_results(pred, lo, params["nms_thesh"], params["num_classes"])
    out = []
    for c in conf_list:
        if isinstance(base, int):
            out.append(0)
            continue
        sel = base[base[:, 5] > np.float32(c)]
        out.append(torch.from_numpy(sel) if sel.shape[0] else 0)
    return out


@pytest.mark.parametrize("train", [False, True])
@pytest.mark.parametrize("name", list(CASES))
def test_get_map_matches_the_reference(monkeypatch, name, train):
    import yolo_amd.harness as harness
    monkeypatch.setattr(harness, "filter_results_multi", oracle_filter_multi)
    _, _, _, _, C, H, confs = CASES[name]
    data = batches(name)
    params = default_params(height=H, num_classes=C)
    params["specific_conf"] = 0.5
    model = ReplayModel(params, [d["pred"] for d in data])
    with contextlib.redirect_stdout(io.StringIO()):
        res = ev.get_map(model, data, train=train, confidence=confs if confs else False)
    check_against_golden(name, res, train)


def test_compute_map_edge_cases():
    params = {"num_classes": 3, "iou_conf": 0.5, "confidence": 0.5,
              "classes": ["a", "b", "c"]}
    box = np.array([[10.0, 10.0, 50.0, 50.0]])
    det = np.array([[10.0, 10.0, 50.0, 50.0, 0.9], [11.0, 10.0, 50.0, 50.0, 0.8]], dtype=np.float32)
    empty = np.array([])
    # class 0: annotation and two detections of it (stored ascending: 0.8 claims first)
    # class 1: detections but no annotation -> AP 0; class 2: annotation, no detection -> AP 0
    all_det = [[[det[::-1], det, empty]]]
    all_ann = [[box, empty, box]]
    mAP, aps = ev.compute_map(all_det, all_ann, 0, None, True, 2, params)
    assert aps[1] == 0 and aps[2] == 0
    # sorted by score: 0.9 is a false positive (its annotation was claimed by 0.8), 0.8 a hit
    assert aps[0] == pytest.approx(0.5)
    assert mAP == pytest.approx(0.25)
    # no detections anywhere
    none = [[[empty, empty, empty]]]
    mAP, aps = ev.compute_map(none, all_ann, 0, None, True, 2, params)
    assert mAP == 0 and aps == [0, 0, 0] or all(a == 0 for a in aps)


def test_annotations_by_class_handles_padding_and_empty_images():
    lab = np.zeros((4, 5))
    lab[0] = [2, 0.5, 0.5, 0.2, 0.4]
    lab[1] = [0, 0.25, 0.25, 0.1, 0.1]
    per, n = ev.annotations_by_class(lab, 416, 3)
    assert n == 2
    assert per[1].shape[0] == 0
    np.testing.assert_allclose(per[2], [[0.4 * 416, 0.3 * 416, 0.6 * 416, 0.7 * 416]])
    per, n = ev.annotations_by_class(np.zeros((4, 5)), 416, 3)
    assert n is None and all(p.shape[0] == 0 for p in per)
