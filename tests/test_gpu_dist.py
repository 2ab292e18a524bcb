"""The multi-process code path on the one GPU a test box has: bench.py with a
single rank forced through torch.distributed (backend "nccl" = RCCL), i.e. RCCL
initialisation on a device, the barriers around the timed region, the
max-over-ranks all-reduce and gather_detections on device tensors.  No scaling
number comes out of this; it shows the N > 1 path executes on hardware."""
import contextlib
import io
import json
import os
import socket
import sys

import pytest
import torch

from helpers import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_bench_runs_its_distributed_path_with_one_rank(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    for k, v in {"YOLO_AMD_BENCH_FORCE_DIST": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1",
                 "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()),
                 "HSA_ENABLE_IPC_MODE_LEGACY": "0"}.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "4",
                                      "--no-cpu-baseline", "--no-pipelined", "--no-traffic", "--no-kernel-pass"])
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    line = [l for l in buf.getvalue().splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["scaling"] == "weak" and res["value"] > 0
    assert res["detections_last_step"] > 0              # gather_detections returned this rank's rows
    assert not torch.distributed.is_initialized()       # the group was torn down again


def test_bench_launches_itself_when_started_without_a_launcher():
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment is how a driver starts
    it: the script then starts its ranks itself (torch.distributed.run as a child process,
    before this process has touched the GPU) and relays rank 0's JSON line.  Run here with one
    rank (--spawn takes the same route for N = 1)."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "YOLO_AMD_BENCH_FORCE_DIST")}
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--steps", "2",
                           "--warmup", "1", "--batch", "4", "--no-cpu-baseline", "--no-pipelined", "--no-traffic",
                           "--no-kernel-pass"], env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    assert "starting 1 rank(s) with torch.distributed.run" in proc.stderr
    line = [l for l in proc.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["value"] > 0 and res["detections_last_step"] > 0
    assert res["roofline_composite"]["bound_images_per_sec_per_gpu"] > res["value"]


def test_gather_detections_on_device_tensors():
    import torch.distributed as dist
    from yolo_amd.dist import gather_detections
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        rows = torch.arange(27, dtype=torch.float32, device="cuda").view(3, 9)
        got = gather_detections(rows, 5)
        want = rows.clone()
        want[:, 0] += 5
        assert torch.equal(got, want) and torch.equal(rows[:, 0], torch.tensor([0., 9., 18.], device="cuda"))
        assert gather_detections(0, 0) == 0
    finally:
        dist.destroy_process_group()
