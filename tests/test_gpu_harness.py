"""GPU parity of the caller post-processing (through the C ABI): the rescale-and-clamp
kernel against the vectors made from the reference's statements and against the
oracle, and the one-NMS confidence sweep against filter_results at every confidence."""
import os

import numpy as np
import pytest
import torch

import gen_inputs as gi
import yolo_oracle as orc
from helpers import assert_bit_equal
from yolo_amd.cfg import default_params

pytestmark = pytest.mark.gpu


def _spin_us(us):
    """Keep torch's current stream busy for about `us` microseconds (the library's idle-wave kernel)."""
    from yolo_amd import _lib
    _lib.check(_lib.lib.yolo_amd_spin(int(us), torch.cuda.current_stream().cuda_stream), "spin")


POST = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_post.npz"))


@pytest.mark.parametrize("name", ["post_416", "post_608_weird", "post_320_one"])
def test_rescale_kernel_matches_the_reference_vectors(name):
    from yolo_amd.harness import rescale_detections
    rows = torch.from_numpy(POST[name + "_rows"]).cuda()
    keep = rows.clone()
    got = rescale_detections(rows, POST[name + "_dims"], int(POST[name + "_height"]))
    assert torch.equal(rows.view(torch.int32), keep.view(torch.int32)), "the input rows were modified"
    assert_bit_equal(got.cpu().numpy(), POST[name + "_out"], name)
    # CPU tensors go through the same kernel and come back on the CPU
    got_cpu = rescale_detections(torch.from_numpy(POST[name + "_rows"]), POST[name + "_dims"],
                                 int(POST[name + "_height"]))
    assert got_cpu.device.type == "cpu"
    assert_bit_equal(got_cpu.numpy(), POST[name + "_out"], name + " (cpu tensor)")
    assert rescale_detections(0, POST[name + "_dims"], 416) == 0


def test_rescale_kernel_large_vs_oracle():
    from yolo_amd.harness import rescale_detections
    r = gi.rs(77)
    K, B, H = 200_000, 32, 416
    rows = r.uniform(-50, 470, (K, 9)).astype(np.float32)
    rows[:, 0] = r.randint(0, B, K)
    dims = np.stack([r.randint(20, 5000, B), r.randint(20, 5000, B)], 1).astype(np.float32)
    got = rescale_detections(torch.from_numpy(rows).cuda(), dims, H).cpu().numpy()
    assert_bit_equal(got, orc.rescale_boxes(rows, dims, H), "200k rows")


@pytest.mark.parametrize("mode,ties,C", [("uniform", 0, 80), ("clustered", 0, 6), ("clustered", 7, 6)])
def test_confidence_sweep_with_one_nms(mode, ties, C):
    from yolo_amd.harness import detections_by_conf, filter_results_multi
    from yolo_amd.utilis import filter_results
    B, N = 3, 1500
    pred_np = gi.nms_pred(41, B, N, C, H=416, mode=mode, tie_levels=ties)
    pred = torch.from_numpy(pred_np).cuda()
    params = default_params(416, C, 3, confidence=0.5, nms_thesh=0.4)
    confs = list(np.arange(start=0.2, stop=0.95, step=0.025)) + [1.01]      # utilis.py:688, plus an empty one
    got = filter_results_multi(pred, params, confs)
    assert len(got) == len(confs)
    want = orc.filter_results_multi(pred_np, confs, 0.4, C)
    for c, g, w in zip(confs, got, want):
        if isinstance(w, int):
            assert isinstance(g, int) and g == 0, f"conf {c}"
        else:
            assert_bit_equal(g.cpu().numpy(), w, f"conf {c:.3f}")
    # and the separate calls of the reference's loop give the same rows
    for c in (confs[0], confs[11], confs[-2]):
        p = dict(params)
        p["confidence"] = float(c)
        one = filter_results(pred, p)
        idx = confs.index(c)
        if isinstance(one, int):
            assert isinstance(got[idx], int)
        else:
            assert torch.equal(one, got[idx])
    assert params["confidence"] == 0.5, "params were modified"
    det = detections_by_conf(pred, params, confs[:3])
    assert len(det) == 3 and len(det[0]) == B and len(det[0][0]) == C
    rows0 = want[0]
    for b in range(B):
        for c in range(C):
            ref = rows0[(rows0[:, 0] == b) & (rows0[:, 7] == c)][:, 1:6]
            got_bc = det[0][b][c]
            assert got_bc.shape[0] == ref.shape[0]
            if ref.shape[0]:
                assert np.all(np.diff(got_bc[:, 4]) >= 0)
                assert_bit_equal(np.sort(got_bc, axis=0), np.sort(ref, axis=0), f"image {b} class {c}")


@pytest.mark.parametrize("depth", [2, 3])
def test_detection_pipeline_returns_what_the_two_calls_return(depth):
    from gpu_helpers import build_model
    from yolo_amd.harness import DetectionPipeline
    from yolo_amd.utilis import filter_results
    model, params, _ = build_model(96)
    params = dict(params)
    params["confidence"], params["nms_thesh"] = 0.3, 0.4
    batches = [torch.from_numpy(gi.images(50 + i, 3, 96)).cuda() for i in range(5)]
    want = []
    for x in batches:
        out = filter_results(model(x), params)
        want.append(out if isinstance(out, int) else out.clone())
    pipe = DetectionPipeline(model, params, depth=depth)
    got = []
    for j, x in enumerate(batches):
        done = pipe.submit(x)
        assert len(done) == (0 if j < depth else 1)      # `depth` batches in flight
        got += [d if isinstance(d, int) else d.clone() for d in done]
    got += [d if isinstance(d, int) else d.clone() for d in pipe.flush()]
    assert pipe.flush() == []
    assert len(got) == len(want)
    for g, w in zip(got, want):
        if isinstance(w, int):
            assert isinstance(g, int) and g == 0
        else:
            assert torch.equal(g.view(torch.int32), w.view(torch.int32))


def test_detection_pipeline_with_a_partial_last_batch_and_unclone_results():
    """Batches of 3, 3, 2, 3 images (a loader's partial batch in the middle and at the
    end): a stream's buffers are replaced on a shape change, and the results in flight at
    that moment must still come back.  Results are NOT cloned here: every output is a
    tensor of its own and must still hold its rows after later submits."""
    from gpu_helpers import build_model
    from yolo_amd.harness import DetectionPipeline
    from yolo_amd.utilis import filter_results
    model, params, _ = build_model(96)
    params = dict(params)
    params["confidence"], params["nms_thesh"] = 0.3, 0.4
    sizes = [3, 3, 2, 3, 3, 3, 2]
    batches = [torch.from_numpy(gi.images(70 + i, b, 96)).cuda() for i, b in enumerate(sizes)]
    want = []
    for x in batches:
        out = filter_results(model(x), params)
        want.append(out if isinstance(out, int) else out.clone())
    assert any(not isinstance(w, int) for w in want)
    pipe = DetectionPipeline(model, params)
    held = []                                   # (index, view) still inside its promised lifetime
    n_checked = 0

    def check(i, g):
        w = want[i]
        if isinstance(w, int):
            assert isinstance(g, int) and g == 0, f"batch {i}"
        else:
            assert not isinstance(g, int), f"batch {i}: detections lost"
            assert torch.equal(g.view(torch.int32), w.view(torch.int32)), f"batch {i}"

    n_out = 0
    for j, x in enumerate(batches):
        done = pipe.submit(x)
        # views returned by the previous submit() are still valid after this one
        for i, g in held:
            check(i, g)
            n_checked += 1
        held = []
        for g in done:
            check(n_out, g)
            held.append((n_out, g))
            n_out += 1
    for g in pipe.flush():
        check(n_out, g)
        n_out += 1
    assert n_out == len(batches)
    assert n_checked == len(batches) - 3


def test_rows_are_ordered_behind_the_launch_stream_for_a_consumer_on_another_stream():
    """filter_results_async on a lane stream, result() on the default stream: the row count
    reaches the host before the kernel that writes the rows has run, so result() has to put
    the consumer's stream behind the lane's last kernel (a device-side event wait) and keep
    the allocator from recycling the rows under it.  The lane is held back by a spin kernel
    so that everything is still queued when result() is called."""
    from yolo_amd import utilis
    from yolo_amd.utilis import filter_results, filter_results_async
    pred_np = gi.nms_pred(61, 4, 3000, 80, H=416, mode="clustered")
    pred = torch.from_numpy(pred_np).cuda()
    params = default_params(416, 80, 3, confidence=0.3, nms_thesh=0.4)
    want = filter_results(pred, params).clone()
    torch.cuda.synchronize()
    lane = torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(lane):
            _spin_us(10000)                 # ~10 ms ahead of the NMS kernels
            pending = filter_results_async(pred, params)
        assert not pending.done()
        rows = pending.result()                            # on the default stream
        got = rows.clone()                                 # a default-stream kernel reading the rows
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int32), want.view(torch.int32))


def test_a_dropped_pending_result_keeps_its_host_counter_out_of_circulation():
    """The scan kernel stores the row count straight into a pinned host word.  A
    PendingDetections dropped before its kernels have run must not hand that word back for
    reuse: it is parked until the launch stream's event has completed."""
    import gc
    from yolo_amd.utilis import _HostCount, filter_results, filter_results_async
    pred = torch.from_numpy(gi.nms_pred(62, 2, 2000, 80, H=416, mode="uniform")).cuda()
    params = default_params(416, 80, 3, confidence=0.4, nms_thesh=0.4)
    filter_results(pred, params)
    torch.cuda.synchronize()
    _HostCount.take().give_back()                         # drains the parked list of earlier tests
    parked = len(_HostCount._limbo)
    lane = torch.cuda.Stream()
    with torch.cuda.stream(lane):
        _spin_us(10000)
        pending = filter_results_async(pred, params)
    host = pending._host
    del pending
    gc.collect()
    assert host in _HostCount._limbo and host not in _HostCount._pool
    assert len(_HostCount._limbo) == parked + 1
    torch.cuda.synchronize()
    assert host.ready(), "the kernel's store arrived in the parked counter"
    other = _HostCount.take()                              # completed counters return to the pool here
    other.give_back()
    assert host not in _HostCount._limbo
    out = filter_results(pred, params)                     # and the path still works
    assert not isinstance(out, int)


@pytest.mark.parametrize("train", [False, True])
@pytest.mark.parametrize("name", ["coco_sweep", "four_class", "single_conf"])
def test_get_map_with_hip_detections_matches_the_reference(name, train):
    """evaluate.get_map end to end on stored predictions: NMS through the library,
    scoring in numpy; every returned number equals what the reference's get_map
    returned on the same data (golden_eval.json)."""
    import contextlib
    import io
    from eval_data import CASES, ReplayModel, batches, check_against_golden
    from yolo_amd import evaluate as ev
    _, _, _, _, C, H, confs = CASES[name]
    data = batches(name)
    params = default_params(height=H, num_classes=C, cuda=True)
    params["specific_conf"] = 0.5
    model = ReplayModel(params, [d["pred"] for d in data], device="cuda")
    with contextlib.redirect_stdout(io.StringIO()):
        res = ev.get_map(model, data, train=train, confidence=confs if confs else False)
    check_against_golden(name, res, train)


def test_concurrent_streams_are_two_distinct_streams_that_overlap():
    """harness.concurrent_streams: a pair of streams that do not share a hardware queue
    (two spin kernels take about as long as one), cached per device."""
    import time
    from yolo_amd.harness import concurrent_streams
    a, b = concurrent_streams(torch.device("cuda", 0))
    assert a.cuda_stream != b.cuda_stream
    assert concurrent_streams("cuda:0") == (a, b)

    def wall(streams, spin=1000):                   # microseconds
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for st in streams:
            with torch.cuda.stream(st):
                _spin_us(spin)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    wall([a, b])
    alone = min(wall([a]) for _ in range(3))
    both = min(wall([a, b]) for _ in range(3))
    assert both < 1.6 * alone, (alone, both)
