"""The drop-in boundary on the host, pinned against the reference itself: the names
the reference's callers import, the Darknet .weights reader (with and without the
cust_train_zero pointer quirk), prep_labels, and prep_params' k-means branch.
Fixtures: tests/golden/golden_{imports,weights,labels}.json, written by the
make_golden_*.py scripts beside them from the imported reference."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import gen_inputs as gi
import yolo_oracle as orc
from helpers import GOLDEN_DIR, ROOT
from yolo_amd.cfg import default_params, write_yolov3_cfg
from yolo_amd.utilis import parse_cfg, prep_labels, prep_params
from yolo_amd.yolo_v3 import yolo_v3


def _golden(name):
    with open(os.path.join(GOLDEN_DIR, name)) as fh:
        return json.load(fh)


def test_every_name_the_callers_import_resolves_through_dropin():
    """detection.py, video.py, get_fps.py and evaluate.py import these names from
    utilis / yolo_v3 / evaluate (golden_imports.json, parsed from the reference);
    with dropin/ first on sys.path each must resolve, and into this package."""
    want = _golden("golden_imports.json")
    attrs = want.pop("model_attributes")
    code = r'''
import importlib, json, sys
sys.path.insert(0, sys.argv[1])
want = json.loads(sys.argv[2])
missing, foreign = [], []
for caller, mods in want.items():
    for mod, names in mods.items():
        m = importlib.import_module(mod)
        if not (getattr(m, "__file__", "") or "").startswith(sys.argv[1]):
            foreign.append((caller, mod, m.__file__))
        for n in names:
            if not hasattr(m, n):
                missing.append((caller, mod, n))
import yolo_v3
for a in json.loads(sys.argv[3]):
    if not hasattr(yolo_v3.yolo_v3, a) and a != "params":
        missing.append(("model", "yolo_v3", a))
print(json.dumps({"missing": missing, "foreign": foreign}))
'''
    out = subprocess.run([sys.executable, "-c", code, os.path.join(ROOT, "dropin"), json.dumps(want),
                          json.dumps(attrs)], capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stderr
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["missing"] == [] and res["foreign"] == []


@pytest.mark.parametrize("zero", [False, True], ids=["plain", "cust_train_zero"])
@pytest.mark.parametrize("case", ["coco_c80_a3", "one_anchor_c4_a1"])
def test_load_weights_reproduces_the_reference_state_dict(case, zero):
    """The same seeded .weights file the reference read for golden_weights.json: every
    tensor of the state_dict must have the digest the reference ended up with --
    with cust_train_zero that includes the layers behind the detection convolutions,
    which read shifted data because the reference skips a constant 255 values."""
    from make_golden_weights import digest_state_dict, weights_file_bytes
    g = _golden("golden_weights.json")["cases"][case]
    C, A = g["C"], g["A"]
    cfg = write_yolov3_cfg(None, C, A)
    table = orc.layer_table(orc.parse_cfg(cfg), 3, C, A)
    with tempfile.NamedTemporaryFile(suffix=".weights", delete=False) as fh:
        fh.write(weights_file_bytes(table, g["seed"]))
    try:
        assert os.path.getsize(fh.name) == g["file_bytes"]
        params = default_params(416, C, A)
        model = yolo_v3(params, parse_cfg(cfg), weight_file=None)
        model.load_weights(fh.name, cust_train_zero=zero)
    finally:
        os.remove(fh.name)
        os.remove(cfg)
    want = g["zero" if zero else "plain"]
    assert int(model.seen) == want["seen"] and model.header.tolist() == want["header"]
    got = digest_state_dict(model.state_dict())
    assert set(got) == set(want["digests"])
    bad = [k for k in got if got[k] != want["digests"][k]]
    assert not bad, f"{len(bad)} tensors differ from the reference's, first: {bad[:3]}"


def test_prep_labels_writes_the_reference_csv():
    from make_golden_labels import run
    want = _golden("golden_labels.json")
    got = run(prep_labels)
    for name in ("all", "sel", "empty"):
        g, w = got[name].splitlines(), want[name].splitlines()
        assert g[0] == w[0], name
        assert sorted(g[1:]) == sorted(w[1:]), name          # glob order is the file system's


def test_prep_params_runs_the_anchor_kmeans_when_the_file_holds_no_list(tmp_path, capsys):
    """/root/reference/utilis.py:680-686: anchors that are not a list are replaced by
    generate_anchor(label_csv, width, height, num_clusters=num_anchors*3)."""
    import random
    from yolo_amd.generate_anchors import generate_anchor
    csv = tmp_path / "labels.csv"
    gi.write_label_csv(str(csv), gi.label_dims(71, 200), seed=71)
    names = tmp_path / "x.names"
    names.write_text("a\nb\n")
    pfile = tmp_path / "params.txt"
    base = {"height": 320, "num_anchors": 1, "anchors": "kmeans", "num_classes": 2}
    pfile.write_text(json.dumps(base))
    cwd = os.getcwd()
    os.chdir(tmp_path)                      # the k-means writes anchors<k>.txt into the cwd
    try:
        random.seed(5)
        p = prep_params(str(pfile), str(csv), {"confidence": 0.3}, names_file=str(names))
        random.seed(5)
        want = generate_anchor(str(csv), 320, 320, num_clusters=3)
    finally:
        os.chdir(cwd)
    assert np.array_equal(np.asarray(p["anchors"]), np.asarray(want)) and np.asarray(p["anchors"]).shape == (3, 2)
    assert p["width"] == 320 and p["confidence"] == 0.3 and p["classes"] == ["a", "b"]
    assert len(p["conf_list"]) == 30
    # a list stays a list turned array
    base["anchors"] = [[10, 20], [30, 40], [50, 60]]
    pfile.write_text(json.dumps(base))
    p = prep_params(str(pfile), str(csv), names_file=str(names))
    assert isinstance(p["anchors"], np.ndarray) and p["anchors"].tolist() == base["anchors"]
