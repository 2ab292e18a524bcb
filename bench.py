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

Rank 0 prints ONE JSON line.  Beside the required fields it carries
  roofline      conv stack (the dominant kernel family) against the dense bf16
                MFMA peak, from HIP events recorded on the launch stream
                inside the timed steps
  cpu_baseline  the CPU port of the same path (torch CPU fp32 forward + the
                reference-shaped Python-loop NMS) on a bounded sample, rank 0,
                N = 1 only
  stages_ms     per-step time of conv stack / decode / NMS
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


def cpu_baseline(sd, H, C, A, conf, nms, budget_s=20.0, max_images=4):
    """CPU port of the path on the host cores: stock torch CPU ops for the
    forward, the reference-shaped per-box Python loop for NMS."""
    import gen_inputs as gi
    import yolo_oracle as orc
    from yolo_amd.cfg import default_params, yolov3_cfg_text
    cores = os.cpu_count() or 1
    # the GPU box gives this job 16 CPUs of a much larger machine
    threads = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    torch.set_num_threads(threads)
    blocks = orc.parse_cfg_text(yolov3_cfg_text(C, A))
    p = default_params(H, C, A)
    x = torch.from_numpy(gi.images(0, 1, H))
    orc.forward(sd, blocks, p, x)                 # warm-up (thread pool, allocator)
    t_fwd = t_nms = 0.0
    done = 0
    t0 = time.perf_counter()
    while done < max_images and (time.perf_counter() - t0) < budget_s:
        x = torch.from_numpy(gi.images(100 + done, 1, H))
        a = time.perf_counter()
        det = orc.forward(sd, blocks, p, x)
        b = time.perf_counter()
        orc.filter_results_torch_loop(det, conf, nms, C)
        c = time.perf_counter()
        t_fwd += b - a
        t_nms += c - b
        done += 1
    total = t_fwd + t_nms
    return {"value": done / total, "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"{done} images at {H}x{H}, batch 1, conf {conf} / NMS {nms}; "
                      f"forward {1e3 * t_fwd / done:.0f} ms/img, NMS {1e3 * t_nms / done:.0f} ms/img"}


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
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # (YOLO_AMD_BENCH_FORCE_DIST=1 runs the multi-process code path with a single rank)
    distributed = world > 1 or os.environ.get("YOLO_AMD_BENCH_FORCE_DIST") == "1"
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import gen_inputs as gi
    from yolo_amd.dist import gather_detections
    from yolo_amd.utilis import filter_results

    H, B, C, A = args.height, args.batch, 80, 3
    model, params, sd, blocks = build_model(H, C, A, device, args.conf, args.nms)
    x = torch.from_numpy(gi.images(rank, B, H)).to(device)
    plan = model._plan(B, H)
    flops_per_batch = sum(2.0 * B * o["Ho"] * o["Wo"] * o["ent"]["cout"] * o["ent"]["cin"] *
                          o["ent"]["k"] ** 2 for o in plan.ops)

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

    for _ in range(args.warmup):
        out = step(False)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    plan.stage_events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(True)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    conv_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in plan.stage_events]))
    dec_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in plan.stage_events]))
    nms_ms = float(np.mean([a.elapsed_time(b) for a, b in nms_events]))
    plan.stage_events = None
    n_rows = 0 if isinstance(out, int) else int(out.shape[0])
    if distributed:
        # outside the timed region: the rows of all ranks in single-process order
        try:
            allrows = gather_detections(out, rank * B, device=device)
            n_rows = 0 if isinstance(allrows, int) else int(allrows.shape[0])
        except Exception as exc:       # the throughput line must not depend on it
            print(f"[bench] gather_detections failed on rank {rank}: {exc}", file=sys.stderr, flush=True)

    pipelined = None
    if world == 1 and not args.no_pipelined:
        from yolo_amd.harness import DetectionPipeline
        pipe = DetectionPipeline(model, params)
        for _ in range(args.warmup):
            pipe.submit(x)
        pipe.flush()
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        for _ in range(args.steps):
            pipe.submit(x)
        pipe.flush()
        torch.cuda.synchronize()
        tp = time.perf_counter() - tp0
        pipelined = {"value": round(B * args.steps / tp, 2), "unit": "images/sec",
                     "ms_per_step": round(1e3 * tp / args.steps, 4),
                     "how": "NMS of batch i on a second HIP stream beside the forward pass of batch i+1"}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * B * args.steps / elapsed
        achieved = flops_per_batch / (conv_ms * 1e-3) / 1e12
        # HBM-side bytes of the conv stack from the PMC passes of tools/pmc_bench.sh over this
        # same command (the profiler cannot run inside the timed region); null when the file is
        # absent or was taken for another batch size
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "round1_conv_traffic.json")
        if os.path.exists(tpath) and (B, H) == (32, 416):
            with open(tpath) as fh:
                traffic = round(json.load(fh)["hbm_bytes_per_launch"])
            traffic_src = "profiles/round1_conv_traffic.json (rocprofv3 --pmc TCC_EA0_RDREQ/WRREQ, bytes per conv launch)"
        n_convs = len(plan.ops) - (1 if getattr(plan, "fused_stem", False) else 0)
        N = plan.N
        result = {
            "metric": "images/sec (node) YOLOv3-416 end-to-end incl. NMS",
            "value": round(value, 2), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"YOLOv3-{H} Darknet-53, batch {B} per GPU, conf {args.conf} / "
                                   f"NMS {args.nms}, seeded synthetic weights and images",
                       "global_batch": world * B, "height": H, "boxes_per_image": N,
                       "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2),
                         "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel": f"conv stack ({n_convs} fused conv launches per step)",
                         "flop_per_launch_avg": flops_per_batch / n_convs,
                         "avg_launch_ms": round(conv_ms / n_convs, 5)},
            "stages_ms": {"conv_stack": round(conv_ms, 4), "decode": round(dec_ms, 4),
                          "nms": round(nms_ms, 4)},
            "nms_us_per_img": round(1e3 * nms_ms / B, 3),
            "decode_gbs": round(2.0 * B * N * (5 + C) * 4 / (dec_ms * 1e-3) / 1e9, 1),
            "detections_last_step": n_rows,
        }
        if pipelined:
            result["pipelined"] = pipelined
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(sd, H, C, A, args.conf, args.nms)
        print(json.dumps(result), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
