"""bench.py's host logic, no GPU: the self-launch route a bare `python bench.py --gpus N`
takes, and the per-layer (composite) roofline arithmetic."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_launches_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus 2` with WORLD_SIZE unset: the script becomes the launcher
    (torch.distributed.run, 127.0.0.1), the two ranks start as fresh processes and, on a box
    without a GPU, fail at the GPU check -- not at an assertion about WORLD_SIZE."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                           "--warmup", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert "starting 2 rank(s) with torch.distributed.run" in proc.stderr
    import torch
    if not torch.cuda.is_available():
        assert proc.returncode != 0
        assert proc.stderr.count("bench.py needs MI355X GPUs") >= 2, proc.stderr[-2000:]
        assert "WORLD_SIZE" not in proc.stderr.split("starting 2 rank(s)")[1].split("Traceback")[0]


def test_a_launcher_with_the_wrong_rank_count_is_reported():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-traffic"],
                          env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode != 0 and "--gpus 2 but the launcher started WORLD_SIZE=3" in proc.stderr


def test_composite_roofline_matches_the_survey_figures():
    """Per-layer max(FLOPs / 2.5 PFLOP/s, bytes / 8 TB/s) at 416 / batch 32: 2.108 TFLOP and
    about 6.2 GB per step, 36 +- 2 of the 75 layers HBM-bound, bound about 1.16 ms = 27.5 k
    images/s against 38.0 k for the MFMA-only bound."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bench
    import yolo_oracle as orc
    from yolo_amd import engine
    from yolo_amd.cfg import yolov3_cfg_text
    table = engine.graph_from_blocks(orc.parse_cfg_text(yolov3_cfg_text()), 3, 80, 3, detection_layers=(81, 93, 105))
    ops, _, _ = engine.build_ops(table, 32, 416, 3)
    rows = bench.composite_roofline(ops, 32)
    assert len(rows) == 75
    assert abs(sum(r["flops"] for r in rows) / 1e12 - 2.1077) < 1e-3
    assert abs(sum(r["bytes"] for r in rows) / 1e9 - 6.2) < 0.1
    t_bound = sum(r["t_bound"] for r in rows)
    t_mfma = sum(r["t_mfma"] for r in rows)
    assert abs(32 / t_mfma - 37957) < 40
    assert 27000 < 32 / t_bound < 28000
    assert 34 <= sum(1 for r in rows if r["bound"] == "hbm") <= 38
