"""Anchor k-means (yolo_amd.generate_anchors) against what the reference's
generate_anchors.py returned on the same seeded label files (golden_anchors.json);
bit-exact, since the arithmetic and its order are the same."""
import contextlib
import io
import json
import os
import random

import numpy as np
import pytest

import gen_inputs as gi
from yolo_amd import generate_anchors as ga

with open(os.path.join(os.path.dirname(__file__), "golden", "golden_anchors.json")) as _fh:
    GOLDEN = json.load(_fh)
CASES = {"nine": (71, 900, 9, 416.0, 416.0), "three": (72, 300, 3, 512.0, 320.0),
         "six": (73, 500, 6, 608.0, 608.0)}


@pytest.mark.parametrize("name", list(CASES))
def test_generate_anchor_matches_the_reference(tmp_path, name):
    seed, n, k, w, h = CASES[name]
    dims = gi.label_dims(seed, n)
    path = str(tmp_path / "label.csv")
    gi.write_label_csv(path, dims, seed)
    random.seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        anchors = ga.generate_anchor(path, w, h, num_clusters=k)
    want = GOLDEN[name]
    assert anchors.tolist() == want["anchors"]
    assert np.all(np.diff(anchors[:, 0]) <= 0)
    cent = anchors / np.array([w, h])
    assert float(ga.avg_IOU(dims, cent)) == want["avg_iou"]
    assert ga.IOU(dims[0], cent).tolist() == want["iou_row0"]
    # rows 2/3 share a width, rows 0/1 are identical: the tie branches
    assert ga.IOU(dims[3], dims[:8]).tolist() == want["iou_row3"]


def test_set_anchors_to_model_hands_groups_to_the_layers_back_to_front(tmp_path, monkeypatch):
    dims = gi.label_dims(74, 200)
    path = str(tmp_path / "label.csv")
    gi.write_label_csv(path, dims, 74)

    class Layer:
        anchors = None

    class Model:
        layer_type_dic = {"yolo": [82, 94, 106]}
        module_list = {82: [Layer()], 94: [Layer()], 106: [Layer()]}
        refolded = 0

        def refold(self):
            self.refolded += 1

    m = Model()
    random.seed(6)
    with contextlib.redirect_stdout(io.StringIO()):
        ga.set_anchors_to_model(m, 2, path, 416, 416)
    random.seed(6)
    with contextlib.redirect_stdout(io.StringIO()):
        want = ga.generate_anchor(path, 416, 416, num_clusters=6)
    assert np.array_equal(m.module_list[106][0].anchors, want[0:2])
    assert np.array_equal(m.module_list[94][0].anchors, want[2:4])
    assert np.array_equal(m.module_list[82][0].anchors, want[4:6])
    assert m.refolded == 1


def test_kmeans_raises_where_the_reference_would_never_return():
    X = np.array([[0.1, 0.1], [0.11, 0.1], [0.5, 0.5], [0.52, 0.5]])
    cent = np.array([[0.1, 0.1], [0.5, 0.5], [0.9, 0.9]])      # the third wins nothing
    with pytest.raises(ValueError, match="lost every sample"):
        ga.kmeans(X, cent, 0.005, "unused", 416, 416)
