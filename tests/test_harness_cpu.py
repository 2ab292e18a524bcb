"""CPU checks of the caller post-processing: the oracle's rescale block against the
vectors torch produced from the reference's statements, and the property the
one-NMS confidence sweep rests on, on the oracle itself."""
import os

import numpy as np
import pytest

import gen_inputs as gi
import yolo_oracle as orc
from helpers import assert_bit_equal

POST = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_post.npz"))
POST_CASES = ["post_416", "post_608_weird", "post_320_one"]


@pytest.mark.parametrize("name", POST_CASES)
def test_oracle_rescale_matches_the_reference_statements(name):
    got = orc.rescale_boxes(POST[name + "_rows"], POST[name + "_dims"], int(POST[name + "_height"]))
    assert_bit_equal(got, POST[name + "_out"], name)
    # untouched columns
    assert_bit_equal(got[:, [0, 5, 6, 7, 8]], POST[name + "_rows"][:, [0, 5, 6, 7, 8]], name + " passthrough")


@pytest.mark.parametrize("mode,ties", [("uniform", 0), ("clustered", 0), ("clustered", 5)])
def test_rows_at_a_higher_confidence_are_a_filter_of_the_rows_at_the_lowest(mode, ties):
    pred = gi.nms_pred(31, 2, 600, 6, H=416, mode=mode, tie_levels=ties)
    confs = [0.05, 0.2, 0.5, 0.7, 0.95, 1.5]
    per_conf = orc.filter_results_multi(pred, confs, 0.4, 6)
    base = per_conf[0]
    assert not isinstance(base, int)
    for c, want in zip(confs, per_conf):
        sel = base[base[:, 5] > np.float32(c)]
        if isinstance(want, int):
            assert sel.shape[0] == 0
        else:
            assert_bit_equal(sel, want, f"conf {c}")
