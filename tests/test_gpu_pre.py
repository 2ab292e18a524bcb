"""GPU parity of the image pre-processing (through the C ABI): the pad / min-max /
channel flip / mean-std part against the vectors made from the reference's
statements (frames already at their letterboxed size, so the resize is the
identity), and the whole transform, resize included, bit for bit against the
oracle's restatement on frames of many sizes."""
import numpy as np
import pytest
import torch

import gen_inputs as gi
import yolo_oracle as orc
from pre_data import CASES, MEAN, MODES, STD, check_against_golden, resized_image

pytestmark = pytest.mark.gpu


def run(frames, side, **kw):
    from yolo_amd.harness import preprocess_frames
    images, dims, canvas = preprocess_frames(frames, side, return_canvas=True, **kw)
    torch.cuda.synchronize()
    return images.cpu().numpy(), dims, canvas.cpu().numpy()


@pytest.mark.parametrize("key", list(MODES))
@pytest.mark.parametrize("name", list(CASES))
def test_pad_and_normalise_match_the_reference_vectors(name, key):
    frame, side = resized_image(name)
    images, dims, canvas = run([frame], side, **{"minmax": False, **MODES[key]})
    assert dims == [(frame.shape[1], frame.shape[0])]
    np.testing.assert_array_equal(canvas[0], orc.letterbox_canvas(frame, side, side).astype(np.uint8))
    check_against_golden(name, key, images[0])


def test_flat_frame_gives_nan_under_min_max():
    images, _, _ = run([np.full((32, 32, 3), 128, dtype=np.uint8)], 32, minmax=True)
    assert np.isnan(images).all()


SIZES = [(1920, 1080, 416), (640, 480, 416), (333, 500, 608), (100, 37, 320), (70, 50, 416),
         (416, 416, 416), (417, 415, 416), (3, 2, 32), (1, 1, 32), (2000, 31, 416),
         (4000, 3000, 416), (17, 4000, 416), (1919, 1081, 608), (5000, 700, 320)]


@pytest.mark.parametrize("iw,ih,side", SIZES)
def test_whole_transform_is_bit_equal_to_the_oracle(iw, ih, side):
    frame = gi.photo(90 + iw % 7, iw, ih)
    want, want_canvas, want_dims = orc.preprocess([frame], side, mean=MEAN, std=STD)
    images, dims, canvas = run([frame], side, mean=MEAN, std=STD)
    np.testing.assert_array_equal(canvas[0], want_canvas[0])
    np.testing.assert_array_equal(images.view(np.int32), want.view(np.int32))
    assert dims == [(iw, ih)]


def test_mixed_batch_over_the_chunk_size_with_pitched_device_frames():
    r = np.random.RandomState(7)
    frames = [gi.photo(200 + k, int(r.randint(20, 900)), int(r.randint(20, 700))) for k in range(37)]
    want, want_canvas, _ = orc.preprocess(frames, 320, minmax=False, div255=True)
    # every other frame is handed over as a view into a wider device image
    given = []
    for k, f in enumerate(frames):
        if k % 2:
            wide = torch.zeros((f.shape[0], f.shape[1] + 5, 3), dtype=torch.uint8, device="cuda")
            wide[:, 2:2 + f.shape[1]] = torch.from_numpy(f).cuda()
            given.append(wide[:, 2:2 + f.shape[1]])
        else:
            given.append(f)
    images, dims, canvas = run(given, 320, minmax=False, div255=True)
    for k in range(len(frames)):
        np.testing.assert_array_equal(canvas[k], want_canvas[k], err_msg=f"frame {k}")
    np.testing.assert_array_equal(images.view(np.int32), want.view(np.int32))


def test_bad_arguments_fail_with_a_message():
    from yolo_amd._lib import YoloAmdError
    from yolo_amd.harness import preprocess_frames
    with pytest.raises(YoloAmdError, match="shrinks to nothing"):
        preprocess_frames([np.zeros((5000, 1, 3), dtype=np.uint8)], 416)
    with pytest.raises(YoloAmdError, match="exclude"):
        preprocess_frames([np.zeros((8, 8, 3), dtype=np.uint8)], 32, minmax=True, div255=True)
    with pytest.raises(ValueError):
        preprocess_frames([np.zeros((8, 8), dtype=np.uint8)], 32)


def test_preprocessed_frames_run_through_the_model_and_rescale():
    """The callers' loop end to end on raw frames: transform, forward, NMS, boxes back
    in each frame's own pixels and inside it."""
    from gpu_helpers import build_model
    from yolo_amd.harness import preprocess_frames, rescale_detections
    from yolo_amd.utilis import filter_results
    model, params, _ = build_model(320, C=4, seed=3, confidence=0.02)
    frames = [gi.photo(300, 500, 375), gi.photo(301, 240, 426)]
    images, dims = preprocess_frames(frames, 320, mean=MEAN, std=STD)
    with torch.no_grad():
        rows = filter_results(model(images), params)
    if not isinstance(rows, int):
        out = rescale_detections(rows, dims, 320).cpu().numpy()
        for r in out:
            w, h = dims[int(r[0])]
            assert 0 <= r[1] <= w and 0 <= r[3] <= w and 0 <= r[2] <= h and 0 <= r[4] <= h
