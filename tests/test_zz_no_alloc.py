"""yolo_amd_filter_results_static (the entry point that allocates nothing: its
workspace is a static device array inside the library) against
yolo_amd_filter_results on the same inputs, through the C ABI: the (K, 9) rows
and K must be identical bit for bit -- on two calls in a row, the second with a
smaller input, so that whatever the first call left in the shared array does not
leak into the second.  Named to be collected last."""
import numpy as np
import pytest
import torch

import gen_inputs as gi
from helpers import assert_bit_equal, c_filter_results

pytestmark = pytest.mark.gpu


def _spin_us(us):
    """Keep torch's current stream busy for about `us` microseconds (the library's idle-wave kernel)."""
    from yolo_amd import _lib
    _lib.check(_lib.lib.yolo_amd_spin(int(us), torch.cuda.current_stream().cuda_stream), "spin")


def _regular(pred, conf, nms, C):
    from yolo_amd import _lib
    from yolo_amd._lib import check, lib
    B, N, D = pred.shape
    rows = torch.full((B * N, 9), float("nan"), dtype=torch.float32, device="cuda")
    cnt = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    nbytes = lib.yolo_amd_filter_results_workspace_bytes(B, N, C)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    check(lib.yolo_amd_filter_results(pred.data_ptr(), B, N, D, C, conf, nms, ws.data_ptr(), nbytes,
                                      rows.data_ptr(), cnt.data_ptr(), _lib.current_stream_ptr(pred.device)),
          "filter_results")
    torch.cuda.synchronize()
    return rows[:int(cnt.item())].cpu().numpy()


def _static(pred, conf, nms, C):
    from yolo_amd import _lib
    from yolo_amd._lib import check, lib
    B, N, D = pred.shape
    rows = torch.full((B * N, 9), float("nan"), dtype=torch.float32, device="cuda")
    cnt = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    check(lib.yolo_amd_filter_results_static(pred.data_ptr(), B, N, D, C, conf, nms, rows.data_ptr(),
                                             cnt.data_ptr(), _lib.current_stream_ptr(pred.device)),
          "filter_results_static")
    torch.cuda.synchronize()
    return rows[:int(cnt.item())].cpu().numpy()


def test_static_scratch_entry_point_matches_the_workspace_one_twice_in_a_row():
    big = gi.nms_pred(seed=81, B=2, N=10647, C=80, mode="clustered")       # 15.9 of the 16 MiB
    small = gi.nms_pred(seed=82, B=1, N=2100, C=4, H=320, mode="clustered", tie_levels=64)
    for pred_np, conf, C in ((big, 0.05, 80), (small, 0.1, 4), (big, 0.5, 80), (small, 1.5, 4)):
        pred = torch.from_numpy(pred_np).cuda()
        got = _static(pred, conf, 0.4, C)
        want = _regular(pred, conf, 0.4, C)
        assert got.shape == want.shape
        if want.shape[0]:
            assert_bit_equal(got, want, f"static vs workspace entry point, conf {conf}")
        assert_bit_equal(want, c_filter_results(pred_np, conf, 0.4, C), "workspace entry point vs oracle")
    assert got.shape[0] == 0                                # the last call kept nothing: K = 0, no rows


def test_static_scratch_entry_point_rejects_what_does_not_fit():
    from yolo_amd._lib import lib
    pred = torch.zeros((1, 22743, 85), dtype=torch.float32, device="cuda")     # a 608 image: 34 MiB
    rows = torch.empty((22743, 9), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    rc = lib.yolo_amd_filter_results_static(pred.data_ptr(), 1, 22743, 85, 80, 0.5, 0.4, rows.data_ptr(),
                                            cnt.data_ptr(), None)
    assert rc < 0 and b"static scratch holds" in lib.yolo_amd_last_error()
    rc = lib.yolo_amd_filter_results_static(pred.data_ptr(), 3, 10647, 85, 80, 0.5, 0.4, rows.data_ptr(),
                                            cnt.data_ptr(), None)
    assert rc < 0


def test_static_scratch_entry_point_refuses_a_second_stream_while_the_first_is_busy():
    """One scratch array per GPU: a call on another stream while the previous call's kernels
    are still queued would corrupt both, so it is refused (-5) and launches nothing; the same
    call succeeds once the first stream has drained, and a call on the SAME stream never waits."""
    from yolo_amd._lib import lib
    pred_np = gi.nms_pred(seed=83, B=1, N=2100, C=4, H=320, mode="clustered")
    pred = torch.from_numpy(pred_np).cuda()
    want = _regular(pred, 0.1, 0.4, 4)
    rows_a = torch.full((2100, 9), float("nan"), dtype=torch.float32, device="cuda")
    rows_b = torch.full((2100, 9), float("nan"), dtype=torch.float32, device="cuda")
    cnt_a = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    cnt_b = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(sa):
        _spin_us(20000)                      # keeps stream A busy for ~20 ms
        rc = lib.yolo_amd_filter_results_static(pred.data_ptr(), 1, 2100, 9, 4, 0.1, 0.4, rows_a.data_ptr(),
                                                cnt_a.data_ptr(), sa.cuda_stream)
        assert rc == 0, lib.yolo_amd_last_error()
        # the same stream again: stream order, no conflict
        rc = lib.yolo_amd_filter_results_static(pred.data_ptr(), 1, 2100, 9, 4, 0.1, 0.4, rows_a.data_ptr(),
                                                cnt_a.data_ptr(), sa.cuda_stream)
        assert rc == 0, lib.yolo_amd_last_error()
    rc = lib.yolo_amd_filter_results_static(pred.data_ptr(), 1, 2100, 9, 4, 0.1, 0.4, rows_b.data_ptr(),
                                            cnt_b.data_ptr(), sb.cuda_stream)
    assert rc == -5 and b"another stream" in lib.yolo_amd_last_error()
    torch.cuda.synchronize()
    assert int(cnt_b.item()) == -1, "the refused call launched something"
    assert_bit_equal(rows_a[:int(cnt_a.item())].cpu().numpy(), want, "stream A's rows")
    rc = lib.yolo_amd_filter_results_static(pred.data_ptr(), 1, 2100, 9, 4, 0.1, 0.4, rows_b.data_ptr(),
                                            cnt_b.data_ptr(), sb.cuda_stream)
    assert rc == 0, lib.yolo_amd_last_error()
    torch.cuda.synchronize()
    assert_bit_equal(rows_b[:int(cnt_b.item())].cpu().numpy(), want, "stream B's rows after A had drained")
