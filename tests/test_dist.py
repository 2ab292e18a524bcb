"""Multi-GPU path on CPU: two gloo ranks each hold a contiguous slice of the
batch, run NMS on it (the oracle stands in for the HIP kernels here), and the
gathered rows must equal the single-process result row for row."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

import gen_inputs as gi
from helpers import c_filter_results


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, empty_rank, result_path):
    import torch.distributed as dist
    from yolo_amd.dist import gather_detections, shard_range
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    pred = gi.nms_pred(seed=71, B=total, N=800, C=6, mode="clustered")
    if empty_rank is not None:
        lo, hi = shard_range(total, empty_rank, world)
        pred[lo:hi, :, 4] = 0.0
    lo, hi = shard_range(total, rank, world)
    rows = c_filter_results(pred[lo:hi], 0.3, 0.4, 6)
    local = torch.from_numpy(rows) if rows.shape[0] else 0
    got = gather_detections(local, lo, device=torch.device("cpu"))
    if rank == 0:
        want = c_filter_results(pred, 0.3, 0.4, 6)
        ok = (isinstance(got, int) and want.shape[0] == 0) or \
            (not isinstance(got, int) and np.array_equal(got.numpy(), want))
        with open(result_path, "w") as fh:
            fh.write("ok" if ok else "mismatch")
    dist.destroy_process_group()


def _run(total, empty_rank, tmp_path):
    result = os.path.join(tmp_path, "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), total, empty_rank, result), nprocs=2, join=True)
    assert open(result).read() == "ok"


def test_shard_range_covers_the_batch():
    from yolo_amd.dist import shard_range
    for total in (1, 5, 8, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_gather_is_shard_invariant(tmp_path):
    _run(5, None, str(tmp_path))


def test_gather_with_an_empty_rank(tmp_path):
    _run(4, 1, str(tmp_path))
