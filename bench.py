"""Headline benchmark: images/sec of YOLOv3-416 inference end to end
(Darknet-53 forward + head decode + confidence-filtered per-class NMS) on N
MI355X GPUs of one node, bf16 MFMA convolutions, batch 32 per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one batch through ``model(images)`` then ``filter_results(prediction,
params)`` (the two calls the reference's callers make), inputs resident in
HBM.  With N > 1 every rank owns its own batch (weak scaling); images are
independent through all three stages, so there is no collective on the data
path: the ranks only meet at the barriers around the timed region and in the
max-over-ranks of the elapsed time.  (yolo_amd.dist.gather_detections, the one
exchange a caller such as an mAP evaluation needs afterwards, is exercised once
after the timed region.)

Started without a launcher (WORLD_SIZE unset) and with --gpus N > 1, this script launches
itself: before anything touches the GPU it starts `python -m torch.distributed.run
--nproc-per-node N bench.py ...` as a child process, relays the child's output and exits
with its status (--spawn forces that route for N = 1 as well).

Rank 0 prints ONE JSON line.  Beside the required fields it carries
  roofline      the dominant kernel (the patch-reuse 3x3 convolution, half of the
                step's GPU time) against the dense bf16 MFMA peak: algorithmic FLOPs
                of its launches / their duration, from HIP events recorded on the
                launch stream around every conv launch of extra passes that follow
                the timed region; `traffic` = HBM-side bytes per launch of that
                kernel from two rocprofv3 --pmc passes over this same script, run as
                child processes before this process touches the GPU (N = 1 only)
  roofline_stack the whole conv stack (74 launches) the same way, its time taken
                from HIP events inside the timed steps
  roofline_composite  the per-layer bound: sum over the conv layers of
                max(FLOPs / MFMA peak, algorithmic bytes / HBM peak), the images/s it
                allows, and the achieved fraction of it (whole step and conv stack alone)
  conv_kernels  per kernel family: launches, time, TFLOP/s (same event passes)
  cpu_baseline  the CPU port of the same path (torch CPU fp32 forward + the
                reference-shaped Python-loop NMS) on a bounded sample, rank 0,
                N = 1 only
  stages_ms     per-step time of the conv stack (head decode fused into the
                detection convolutions) and of the NMS
  nms           us per image and achieved GB/s on the algorithmic bytes (the
                prediction tensor read once, B*N*D*4)
  pipelined     (N = 1) the same K batches through yolo_amd.harness.DetectionPipeline:
                NMS of batch i on a second HIP stream beside the forward pass of
                batch i+1; reported beside `value`, which stays the plain two-call loop
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"),
           os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0     # MI355X dense bf16 (2:1 sparsity excluded)
HBM_PEAK_GBS = 8000.0


def build_model(H, C, A, device, conf, nms):
    import gen_inputs as gi
    from yolo_amd.cfg import default_params, write_yolov3_cfg
    from yolo_amd.engine import graph_from_blocks
    from yolo_amd.utilis import parse_cfg
    from yolo_amd.yolo_v3 import yolo_v3
    cfg = write_yolov3_cfg(None, C, A)
    blocks = parse_cfg(cfg)
    os.remove(cfg)
    params = default_params(H, C, A, cuda=False, confidence=conf, nms_thesh=nms)
    model = yolo_v3(params, blocks, weight_file=None)
    # the table synthetic_state_dict wants has the keys cin/cout/k/bn/stride
    table = graph_from_blocks(blocks, 3, C, A)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in gi.synthetic_state_dict(table, 3).items()}
    model.load_state_dict(sd)
    params["cuda"] = True
    model.to(device)
    model.eval()
    return model, params, sd, blocks


def cpu_baseline(sd, H, C, A, conf, nms, budget_s=24.0):
    """CPU port of the path on the host cores: stock torch CPU ops for the forward, the
    reference-shaped per-box Python loop for NMS, all host threads this job owns.  Batch 1
    (3 warm-up + up to 10 timed passes) and batch 32 (the GPU workload's batch; one pass
    takes seconds, so 1 warm-up + as many timed passes as the budget allows, at least 2):
    median and minimum wall-clock per pass."""
    import gen_inputs as gi
    import yolo_oracle as orc
    from yolo_amd.cfg import default_params, yolov3_cfg_text
    cores = os.cpu_count() or 1
    # the GPU box gives this job 16 CPUs of a much larger machine
    threads = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    torch.set_num_threads(threads)
    blocks = orc.parse_cfg_text(yolov3_cfg_text(C, A))
    p = default_params(H, C, A)

    def one_pass(x):
        a = time.perf_counter()
        det = orc.forward(sd, blocks, p, x)
        b = time.perf_counter()
        orc.filter_results_torch_loop(det, conf, nms, C)
        c = time.perf_counter()
        return b - a, c - b

    def run(batch, warm, max_timed, min_timed, budget):
        t0 = time.perf_counter()
        xs = [torch.from_numpy(gi.images(100 + 7 * batch + i, batch, H)) for i in range(2)]
        for i in range(warm):
            one_pass(xs[i % 2])
        fwd, nm = [], []
        while len(fwd) < max_timed and (len(fwd) < min_timed or time.perf_counter() - t0 < budget):
            f, n = one_pass(xs[len(fwd) % 2])
            fwd.append(f)
            nm.append(n)
        tot = np.asarray(fwd) + np.asarray(nm)
        return {"batch": batch, "warmup": warm, "timed": len(fwd),
                "images_per_sec_median": round(batch / float(np.median(tot)), 3),
                "images_per_sec_best": round(batch / float(tot.min()), 3),
                "ms_per_pass_median": round(1e3 * float(np.median(tot)), 1),
                "ms_per_pass_min": round(1e3 * float(tot.min()), 1),
                "forward_ms_per_img_median": round(1e3 * float(np.median(fwd)) / batch, 1),
                "nms_ms_per_img_median": round(1e3 * float(np.median(nm)) / batch, 1)}

    b1 = run(1, 3, 10, 3, budget_s * 0.25)
    b32 = run(32, 1, 10, 2, budget_s * 0.75)
    return {"value": b32["images_per_sec_median"], "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"{H}x{H}, conf {conf} / NMS {nms}: batch 32 ({b32['warmup']} warm-up + {b32['timed']} timed "
                      f"passes, median) and batch 1 ({b1['warmup']} + {b1['timed']}); value = batch 32 median",
            "batch32": b32, "batch1": b1}


def composite_roofline(ops, B):
    """Per conv LAYER: algorithmic FLOPs and HBM bytes (every layer reads its input once --
    the fp32 image for the first, bf16 after, an upsampled source at its own size --, its
    residual operand once, its weights once per batch, and writes its output once, bf16 or
    the fp32 prediction rows), and max(FLOPs / MFMA peak, bytes / HBM peak) seconds."""
    rows = []
    for op in ops:
        e = op["ent"]
        fl = 2.0 * B * op["Ho"] * op["Wo"] * e["cout"] * e["cin"] * e["k"] ** 2
        inb = sum(B * (op["H"] >> up) * (op["W"] >> up) * c * (4 if op["first"] else 2) for _, c, up in op["src"])
        outb = B * op["Ho"] * op["Wo"] * e["cout"] * (4 if op["fp32"] else 2)
        by = inb + outb + (outb if op["res"] else 0) + e["cout"] * e["cin"] * e["k"] ** 2 * 2
        t_m, t_h = fl / (MFMA_BF16_DENSE_PEAK_TFLOPS * 1e12), by / (HBM_PEAK_GBS * 1e9)
        rows.append({"index": op["index"], "k": e["k"], "stride": e["stride"], "cin": e["cin"], "cout": e["cout"],
                     "Ho": op["Ho"], "flops": fl, "bytes": float(by), "t_mfma": t_m, "t_hbm": t_h,
                     "bound": "mfma" if t_m >= t_h else "hbm", "t_bound": max(t_m, t_h)})
    return rows


def self_launch(args, argv):
    """No launcher started this process and more than one GPU is asked for (or --spawn):
    become the launcher.  Nothing here touches the GPU; the ranks are fresh processes."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    child_args = [a for a in argv if a != "--spawn"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *child_args]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print(f"[bench] no launcher in the environment: starting {args.gpus} rank(s) with torch.distributed.run",
          file=sys.stderr, flush=True)
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def measure_traffic(argv_tail, steps=3, warmup=1, timeout_s=240):
    """HBM-side bytes of the conv kernels from rocprofv3 PMC passes over this script
    (reads and writes in separate passes: the TCC block holds four counters).  Run as
    child processes, each with a time limit, before the parent initialises the GPU.
    Returns (per-kernel-family dict, None) or (None, reason)."""
    import collections
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="yolo_amd_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(int)
    try:
        for name, counters in (("rd", ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"]),
                               ("wr", ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"])):
            cmd = [rocprof, "--pmc", *counters, "--output-format", "csv", "-d", os.path.join(tmp, name), "--",
                   sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", str(warmup),
                   "--no-cpu-baseline", "--no-pipelined", "--no-traffic", "--no-kernel-pass", *argv_tail]
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                                    start_new_session=True)
            try:
                _, err = proc.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)
                proc.wait()
                return None, f"rocprofv3 pass {name} exceeded {timeout_s} s"
            if proc.returncode != 0:
                return None, f"rocprofv3 pass {name} failed ({proc.returncode}): {err.decode(errors='replace')[-200:]}"
            for f in glob.glob(os.path.join(tmp, name, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        kn = r["Kernel_Name"]
                        if "yolo_amd::conv" not in kn:
                            continue
                        fam = kn.split("(")[0].replace("void ", "").replace("yolo_amd::", "")
                        tot[fam][r["Counter_Name"]] += float(r["Counter_Value"])
                        if r["Counter_Name"] == counters[0] and name == "rd":
                            launches[fam] += 1
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for fam, c in tot.items():
        rd, rd32 = c["TCC_EA0_RDREQ_sum"], c["TCC_EA0_RDREQ_32B_sum"]
        wr, wr64 = c["TCC_EA0_WRREQ_sum"], c["TCC_EA0_WRREQ_64B_sum"]
        # gfx950 tallies the 128-byte requests of 16-byte-per-lane loads once: 128 B per
        # non-32B read request; writes are 64 B or 32 B per request
        n = max(launches[fam], 1)
        out[fam] = {"launches_profiled": launches[fam],
                    "read_bytes_per_launch": ((rd - rd32) * 128 + rd32 * 32) / n,
                    "write_bytes_per_launch": (wr64 * 64 + (wr - wr64) * 32) / n}
    if not out:
        return None, "no conv kernel in the counter files"
    return out, None


def gemm_calibration(dev, n=8192, reps=10):
    """What the vendor GEMM library (torch.matmul: hipBLASLt / rocBLAS) sustains on THIS GPU for a large square
    bf16 product of random data, after the timed region: a yardstick for the MFMA fractions above -- the nominal
    peak they are priced against is not reached by any bf16 kernel at the board's power limit.  Not part of the path."""
    a = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
    b = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(a, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        torch.matmul(a, b)
    e1.record()
    torch.cuda.synchronize()
    tf = 2.0 * n ** 3 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12
    return {"what": f"torch.matmul {n}^3 bf16, random data, vendor GEMM library", "tflops": round(tf, 1),
            "frac_of_nominal_peak": round(tf / MFMA_BF16_DENSE_PEAK_TFLOPS, 4)}


KERNEL_FAMILY = {"patch": "conv3x3_patch_kernel", "stem": "conv_stem_kernel", "thin": "conv3x3_thin_kernel",
                 "first": "conv_first_kernel", "block": "conv_block64_kernel"}


def kernel_pass(plan, passes):
    """HIP events around every conv launch (on the stream the kernels run on), `passes`
    times over the whole stack, on the data the last step left in the arena.  Returns
    {family: {"launches", "ms", "flops"}}, averaged over the passes."""
    import ctypes
    from yolo_amd import _lib
    from yolo_amd._lib import check, lib
    groups = []                                   # (first desc, count, family, flops)
    j = 0
    while j < len(plan.ops):
        n = 1 + int(plan.descs[j].fuse_next)      # a fused group is one launch (the stem: up to three layers)
        ch = _lib.conv_choice(plan.descs[j])
        fam = KERNEL_FAMILY.get(ch["kernel"]) or \
            f"conv_igemm_kernel<{ch['bm']}, {ch['bn']}, {ch['bk']}, act {ch['act_ksteps']}>"
        if ch["kernel"] == "patch" and ch["patch_flat"]:
            fam = "conv3x3_flat_kernel"               # the patch-reuse kernel's default form (flat patch planes)
        fl = sum(2.0 * plan.B * o["Ho"] * o["Wo"] * o["ent"]["cout"] * o["ent"]["cin"] * o["ent"]["k"] ** 2
                 for o in plan.ops[j:j + n])
        groups.append((j, n, fam, fl))
        j += n
    stream = _lib.current_stream_ptr(plan.device)
    size = ctypes.sizeof(_lib.ConvDesc)
    base = ctypes.addressof(plan.descs)
    acc = {}
    per_launch = [{"first_op": j0, "n_ops": n, "family": fam, "flops": fl, "ms": 0.0} for j0, n, fam, fl in groups]
    for _ in range(passes):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(groups) + 1)]
        evs[0].record()
        for g, (j0, n, fam, fl) in enumerate(groups):
            ptr = ctypes.cast(base + j0 * size, ctypes.POINTER(_lib.ConvDesc))
            check(lib.yolo_amd_conv_forward(ptr, n, plan.zero_page.data_ptr(), stream), "conv_forward")
            evs[g + 1].record()
        torch.cuda.synchronize()
        for g, (j0, n, fam, fl) in enumerate(groups):
            a = acc.setdefault(fam, {"launches": 0, "ms": 0.0, "flops": 0.0})
            a["launches"] += 1
            dt = evs[g].elapsed_time(evs[g + 1])
            a["ms"] += dt
            a["flops"] += fl
            per_launch[g]["ms"] += dt / passes
    for a in acc.values():
        a["launches"] //= passes
        a["ms"] /= passes
        a["flops"] /= passes
    return acc, per_launch


def write_layer_table(path, plan, layers, per_launch, scale, B, H):
    """One row per conv launch: what ran, its time (HIP events on the launch stream, scaled
    like conv_kernels), algorithmic FLOPs and bytes, the bound that applies and the fraction
    of it that was reached."""
    by_index = {op["index"]: r for op, r in zip(plan.ops, layers)}
    lines = [f"# per-launch roofline, YOLOv3-{H}, batch {B}: time from HIP events around every conv launch "
             f"(scaled by {scale:.3f} to the stack's time inside the timed steps)",
             f"# bound = max(FLOP / 2.5 PFLOP/s, algorithmic bytes / 8 TB/s); frac = bound time / measured time",
             f"{'block':>5s} {'k':>1s} {'s':>1s} {'cin':>5s} {'cout':>5s} {'map':>4s} {'kernel':44s} {'us':>7s} "
             f"{'GFLOP':>7s} {'MB':>7s} {'TFLOP/s':>8s} {'GB/s':>7s} {'bound':>5s} {'bound us':>8s} {'frac':>5s}"]
    tot_us = tot_bound = 0.0
    for g in per_launch:
        ops = plan.ops[g["first_op"]:g["first_op"] + g["n_ops"]]
        rows = [by_index[o["index"]] for o in ops]
        fl, by = sum(r["flops"] for r in rows), sum(r["bytes"] for r in rows)
        fam_fused = g["family"].startswith(("conv_stem", "conv_block"))
        if g["n_ops"] == 2 and not fam_fused:   # a 1x1 layer in the epilogue of the layer in front: that
            by -= B * ops[0]["Ho"] * ops[0]["Wo"] * ops[0]["ent"]["cout"] * 2     # layer's output is written, not re-read
        elif g["n_ops"] >= 2:     # a fused group: the first of the layers' output never goes to memory
            inter = B * ops[0]["Ho"] * ops[0]["Wo"] * ops[0]["ent"]["cout"] * 2
            by -= 2 * inter
            if g["n_ops"] == 3:   # ... and the third layer takes the second one's output from LDS (it is still written)
                by -= B * ops[1]["Ho"] * ops[1]["Wo"] * ops[1]["ent"]["cout"] * 2
            if g["family"].startswith("conv_block"):   # ... and the block's input doubles as the shortcut operand
                by -= B * ops[1]["Ho"] * ops[1]["Wo"] * ops[1]["ent"]["cout"] * 2
        tb = max(fl / (MFMA_BF16_DENSE_PEAK_TFLOPS * 1e12), by / (HBM_PEAK_GBS * 1e9))
        bound = "mfma" if fl / (MFMA_BF16_DENSE_PEAK_TFLOPS * 1e12) >= by / (HBM_PEAK_GBS * 1e9) else "hbm"
        us = g["ms"] * scale * 1e3
        last = rows[-1]
        blk = "+".join(str(o["index"]) for o in ops)
        lines.append(f"{blk:>5s} {last['k']:1d} {last['stride']:1d} {last['cin']:5d} {last['cout']:5d} {last['Ho']:4d} "
                     f"{g['family'][:44]:44s} {us:7.1f} {fl / 1e9:7.2f} {by / 1e6:7.1f} {fl / us / 1e6:8.0f} "
                     f"{by / us / 1e3:7.0f} {bound:>5s} {tb * 1e6:8.1f} {tb * 1e6 / us:5.2f}")
        tot_us += us
        tot_bound += tb * 1e6
    lines.append(f"# total {tot_us:.1f} us measured, {tot_bound:.1f} us bound, fraction {tot_bound / tot_us:.3f}")
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--height", type=int, default=416)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--conf", type=float, default=0.5)
    ap.add_argument("--nms", type=float, default=0.4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the extra pipelined measurement")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC child passes")
    ap.add_argument("--no-calibration", action="store_true", help="skip the vendor-GEMM yardstick")
    ap.add_argument("--no-kernel-pass", action="store_true", help="skip the per-launch event passes")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="batches in flight per GPU, each on a HIP stream of its own (1: one batch after the other)")
    ap.add_argument("--spawn", action="store_true",
                    help="start the rank(s) through torch.distributed.run even for --gpus 1")
    ap.add_argument("--layer-table", default=None,
                    help="write the per-launch roofline table (time, FLOP, bytes, bound, fraction) to this file")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        sys.exit(self_launch(args, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # PMC passes first: they run as child processes and must be over before this
    # process initialises the GPU
    traffic, traffic_note = None, "not measured"
    if world == 1 and rank == 0 and not args.no_traffic:
        tail = ["--height", str(args.height), "--batch", str(args.batch), "--conf", str(args.conf),
                "--nms", str(args.nms)]
        traffic, traffic_note = measure_traffic(tail)
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # (YOLO_AMD_BENCH_FORCE_DIST=1 runs the multi-process code path with a single rank)
    distributed = world > 1 or os.environ.get("YOLO_AMD_BENCH_FORCE_DIST") == "1"
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    assert world == args.gpus, f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks"

    import gen_inputs as gi
    from yolo_amd.dist import gather_detections
    from yolo_amd.utilis import filter_results

    H, B, C, A = args.height, args.batch, 80, 3
    model, params, sd, blocks = build_model(H, C, A, device, args.conf, args.nms)
    x = torch.from_numpy(gi.images(rank, B, H)).to(device)
    plan = model._plan(B, H)
    flops_per_batch = sum(2.0 * B * o["Ho"] * o["Wo"] * o["ent"]["cout"] * o["ent"]["cin"] *
                          o["ent"]["k"] ** 2 for o in plan.ops)

    from yolo_amd.utilis import filter_results_async
    F = max(1, args.in_flight)
    from yolo_amd.harness import concurrent_streams
    # (streams that do not share a hardware queue: two that do would run one after the other)
    lanes = list(concurrent_streams(device, F)) if F > 1 else [torch.cuda.Stream(device=device)]
    xs = [x] + [torch.from_numpy(gi.images(1000 + rank * F + s, B, H)).to(device) for s in range(1, F)]
    pending = [None] * F

    def flight_step(i):
        """One pass of the hot path over one batch: model(x) -> filter_results, queued on
        stream i % F behind that stream's previous batch (whose rows are collected first);
        the batches of the other streams are in flight meanwhile."""
        s = i % F
        done = pending[s].result() if pending[s] is not None else None
        with torch.cuda.stream(lanes[s]):
            pending[s] = filter_results_async(model(xs[s]), params)
        return done

    def drain():
        outs = []
        for s in range(F):
            if pending[s] is not None:
                outs.append(pending[s].result())
                pending[s] = None
        return outs

    nms_events = []

    def step(record):
        pred = model(x)
        if record:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        out = filter_results(pred, params)
        if record:
            e1.record()
            nms_events.append((e0, e1))
        return out

    for i in range(max(args.warmup, F)):
        flight_step(i)
    drain()
    for _ in range(args.warmup):
        out = step(False)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    # ---- the timed region: exactly `steps` passes, F batches in flight
    t0 = time.perf_counter()
    for i in range(args.steps):
        flight_step(i)
    outs = drain()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # ---- the same passes one after the other on one stream, with events around the stages
    plan.stage_events = []
    ts0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(True)
    torch.cuda.synchronize()
    serial_elapsed = time.perf_counter() - ts0

    conv_ms = float(np.mean([e[0].elapsed_time(e[2]) for e in plan.stage_events]))
    nms_ms = float(np.mean([a.elapsed_time(b) for a, b in nms_events]))
    plan.stage_events = None
    kernels = per_launch = None
    if rank == 0 and not args.no_kernel_pass:
        kernels, per_launch = kernel_pass(plan, max(3, min(args.steps, 10)))
    n_rows = 0 if isinstance(out, int) else int(out.shape[0])
    if distributed:
        # outside the timed region: the rows of all ranks in single-process order
        try:
            allrows = gather_detections(out, rank * B, device=device)
            n_rows = 0 if isinstance(allrows, int) else int(allrows.shape[0])
        except Exception as exc:       # the throughput line must not depend on it
            print(f"[bench] gather_detections failed on rank {rank}: {exc}", file=sys.stderr, flush=True)

    if F > 1 and outs:
        out = outs[-1]
    pipelined = None
    if world == 1 and not args.no_pipelined:
        from yolo_amd.harness import DetectionPipeline
        pipe = DetectionPipeline(model, params)
        for i in range(args.warmup):
            pipe.submit(xs[i % F])
        pipe.flush()
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        for i in range(args.steps):
            pipe.submit(xs[i % F])
        pipe.flush()
        torch.cuda.synchronize()
        tp = time.perf_counter() - tp0
        pipelined = {"value": round(B * args.steps / tp, 2), "unit": "images/sec",
                     "ms_per_step": round(1e3 * tp / args.steps, 4),
                     "how": "harness.DetectionPipeline: the callers' loop with two batches in flight on two "
                            "HIP streams, results handed back in order"}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * B * args.steps / elapsed
        achieved = flops_per_batch / (conv_ms * 1e-3) / 1e12
        n_convs = len(plan.ops) - sum(int(d.fuse_next) for d in plan.descs)      # launches: a fused group is one
        N = plan.N
        stack = {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_BF16_DENSE_PEAK_TFLOPS,
                 "unit": "TFLOP/s", "frac": round(achieved / MFMA_BF16_DENSE_PEAK_TFLOPS, 4),
                 "kernel": f"conv stack ({n_convs} fused conv launches per step, head decode included)",
                 "flop_per_launch_avg": flops_per_batch / n_convs, "avg_launch_ms": round(conv_ms / n_convs, 5),
                 "traffic": None}
        if traffic:
            n_prof = max(1, sum(t["launches_profiled"] for t in traffic.values()))
            stack["traffic"] = round(sum((t["read_bytes_per_launch"] + t["write_bytes_per_launch"]) *
                                         t["launches_profiled"] for t in traffic.values()) / n_prof)
        roofline, conv_kernels, scale = dict(stack), None, 1.0
        if kernels:
            # An event between two launches costs about 3 us of stream time, so the
            # per-launch intervals sum to more than the stack takes inside the timed
            # steps: scale them to that time (the raw figure stays in the line).
            raw_total = sum(k["ms"] for k in kernels.values())
            scale = conv_ms / raw_total if raw_total > 0 else 1.0
            for k in kernels.values():
                k["raw_ms"] = k["ms"]
                k["ms"] *= scale
            conv_kernels = {}
            for fam, a in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"]):
                conv_kernels[fam] = {"launches_per_step": a["launches"], "ms_per_step": round(a["ms"], 4),
                                     "avg_launch_ms": round(a["ms"] / a["launches"], 5),
                                     "tflops": round(a["flops"] / (a["ms"] * 1e-3) / 1e12, 1)}
            dom, a = max(kernels.items(), key=lambda kv: kv[1]["ms"])
            ach = a["flops"] / (a["ms"] * 1e-3) / 1e12
            tr = None
            if traffic and "<" not in dom:         # (the im2col instantiations share one name stem)
                tkeys = [k for k in traffic if k.startswith(dom)]      # every instantiation of the family
                if tkeys:
                    n_fam = sum(traffic[k]["launches_profiled"] for k in tkeys)
                    tr = round(sum((traffic[k]["read_bytes_per_launch"] + traffic[k]["write_bytes_per_launch"]) *
                                   traffic[k]["launches_profiled"] for k in tkeys) / max(1, n_fam))
            roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_BF16_DENSE_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), "traffic": tr,
                        "kernel": f"{dom} ({a['launches']} launches per step, "
                                  f"{100 * a['ms'] / sum(k['ms'] for k in kernels.values()):.0f} % of the conv time)",
                        "flop_per_launch_avg": a["flops"] / a["launches"],
                        "avg_launch_ms": round(a["ms"] / a["launches"], 5),
                        "avg_launch_ms_between_events": round(a["raw_ms"] / a["launches"], 5),
                        "how": "HIP events around every conv launch, on the launch stream, in extra passes "
                               "after the timed region; intervals scaled by "
                               f"{scale:.3f} so that they sum to the conv stack's time inside the timed steps "
                               "(an event between two launches costs about 3 us)"}
        roofline["traffic_note"] = traffic_note if not traffic else \
            "rocprofv3 --pmc TCC_EA0_RDREQ/_32B and TCC_EA0_WRREQ/_64B, two child passes over this script; " \
            "128 B per non-32B read request, 64/32 B per write request; bytes per launch of this kernel"
        # ---- the per-layer bound (MFMA for the 3x3 layers, HBM for most 1x1 layers)
        layers = composite_roofline(plan.ops, B)
        t_bound = sum(r["t_bound"] for r in layers)
        t_mfma_only = sum(r["t_mfma"] for r in layers)
        composite = {"bound": "per layer max(FLOPs / 2.5 PFLOP/s, algorithmic bytes / 8 TB/s), summed over "
                              f"the {len(layers)} conv layers",
                     "layers_mfma_bound": sum(1 for r in layers if r["bound"] == "mfma"),
                     "layers_hbm_bound": sum(1 for r in layers if r["bound"] == "hbm"),
                     "algorithmic_gbytes_per_step": round(sum(r["bytes"] for r in layers) / 1e9, 3),
                     "bound_ms_per_step": round(1e3 * t_bound, 4),
                     "bound_images_per_sec_per_gpu": round(B / t_bound, 1),
                     "mfma_only_bound_images_per_sec_per_gpu": round(B / t_mfma_only, 1),
                     "frac_of_composite": round(value / world / (B / t_bound), 4),
                     "frac_of_mfma_only": round(value / world / (B / t_mfma_only), 4),
                     "conv_stack_serial_frac_of_composite": round(1e3 * t_bound / conv_ms, 4)}
        if per_launch and args.layer_table:
            write_layer_table(args.layer_table, plan, layers, per_launch, scale, B, H)
        pred_bytes = float(B) * N * (5 + C) * 4
        result = {
            "metric": "images/sec (node) YOLOv3-416 end-to-end incl. NMS",
            "value": round(value, 2), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"YOLOv3-{H} Darknet-53, batch {B} per GPU, conf {args.conf} / "
                                   f"NMS {args.nms}, seeded synthetic weights and images",
                       "global_batch": world * B, "height": H, "boxes_per_image": N,
                       "parallelism": f"dp{world}", "batches_in_flight_per_gpu": F},
            "serial": {"value": round(B * args.steps / serial_elapsed, 2), "unit": "images/sec (this rank)",
                       "ms_per_step": round(1e3 * serial_elapsed / args.steps, 4),
                       "how": "the same steps one batch after the other on one stream (what stages_ms, "
                              "roofline_stack and the per-kernel figures are measured in)"},
            "roofline": roofline,
            "roofline_stack": stack,
            "roofline_composite": composite,
            "stages_ms": {"conv_stack": round(conv_ms, 4), "nms": round(nms_ms, 4)},
            "nms": {"us_per_img": round(1e3 * nms_ms / B, 3),
                    "achieved_gbs": round(pred_bytes / (nms_ms * 1e-3) / 1e9, 1), "peak_gbs": HBM_PEAK_GBS,
                    "frac": round(pred_bytes / (nms_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "bytes": "the (B, N, 5+C) fp32 prediction read once"},
            "nms_us_per_img": round(1e3 * nms_ms / B, 3),
            "detections_last_step": n_rows,
        }
        if conv_kernels:
            result["conv_kernels"] = conv_kernels
        if traffic:
            result["hbm_traffic_per_launch"] = {k: {kk: round(vv) for kk, vv in v.items()} for k, v in traffic.items()}
        if pipelined:
            result["pipelined"] = pipelined
        if world == 1 and not args.no_calibration:
            result["calibration"] = gemm_calibration(device)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(sd, H, C, A, args.conf, args.nms)
        print(json.dumps(result), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
