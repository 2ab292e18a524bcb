"""The oracle's image transforms: pad / min-max / channel flip / mean-std against
what numpy and torch returned for the reference's statements (golden_pre.npz), and
the restated cubic resize against properties and an independent float bicubic
(torch's, same kernel constant and pixel mapping) -- its parity with cv2 itself is
unpinned: cv2 is not installed and the reference holds no resized image."""
import numpy as np
import pytest
import torch

import gen_inputs as gi
import yolo_oracle as orc
from pre_data import CASES, MODES, PRE, check_against_golden, resized_image


@pytest.mark.parametrize("key", list(MODES))
@pytest.mark.parametrize("name", list(CASES))
def test_oracle_pad_and_normalise_match_the_reference_statements(name, key):
    resized, side = resized_image(name)
    canvas = orc.letterbox_canvas(resized, side, side)
    check_against_golden(name, key, orc.canvas_to_input(canvas, **MODES[key]))


def test_flat_image_normalises_to_nan_like_numpy_does():
    flat = np.full((32, 32, 3), 128, dtype=np.uint8)
    got = orc.canvas_to_input(orc.letterbox_canvas(flat, 32, 32), minmax=True)
    assert np.isnan(got).all() and np.isnan(PRE["flat_32_minmax"]).all()


@pytest.mark.parametrize("iw,ih,side", [(1920, 1080, 416), (640, 480, 416), (333, 500, 608),
                                        (416, 416, 416), (100, 37, 320), (17, 4000, 416)])
def test_letterbox_geometry_fits_and_centres(iw, ih, side):
    new_w, new_h, left, top = orc.letterbox_geometry(iw, ih, side, side)
    assert 1 <= new_w <= side and 1 <= new_h <= side and max(new_w, new_h) >= side - 1
    assert left == (side - new_w) // 2 and top == (side - new_h) // 2


def test_cubic_resize_same_size_is_the_identity_and_constant_stays_constant():
    img = gi.photo(85, 70, 50)
    assert np.array_equal(orc.resize_cubic_u8(img, 70, 50), img)
    const = np.full((9, 13, 3), 77, dtype=np.uint8)
    assert (orc.resize_cubic_u8(const, 40, 31) == 77).all()
    assert (orc.resize_cubic_u8(const, 5, 3) == 77).all()


@pytest.mark.parametrize("iw,ih,nw,nh", [(300, 200, 416, 277), (1280, 720, 416, 234),
                                         (97, 211, 191, 416), (64, 64, 63, 65)])
def test_cubic_resize_agrees_with_a_float_bicubic_to_one_level(iw, ih, nw, nh):
    img = gi.photo(86, iw, ih)
    got = orc.resize_cubic_u8(img, nw, nh).astype(np.int64)
    t = torch.from_numpy(img).permute(2, 0, 1)[None].double()
    ref = torch.nn.functional.interpolate(t, size=(nh, nw), mode="bicubic", align_corners=False)
    ref = ref[0].permute(1, 2, 0).numpy()
    want = np.clip(np.rint(ref), 0, 255).astype(np.int64)
    diff = np.abs(got - want)
    # the 11-bit weights move a value by a fraction of a level: differences only
    # where the float result sits next to a rounding boundary
    assert diff.max() <= 1
    assert (diff > 0).mean() < 0.06
