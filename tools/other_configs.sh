#!/bin/bash
# The other reference configurations (608 batch 32, NMS stress, batch-1 latency cases), for orientation.
cd "$GRAFT_REPO_ROOT"
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-traffic --no-pipelined "$@" > gpurun_out/oc.json 2> gpurun_out/oc.err || { tail -3 gpurun_out/oc.err; exit 1; }
  python - "$*" <<'PY'
import json, sys
r = json.load(open("gpurun_out/oc.json"))
print(sys.argv[1], "->", r["value"], "images/s,", r["ms_per_step"], "ms/step; serial", r["serial"]["value"], r["serial"]["ms_per_step"],
      "stages", r["stages_ms"], "stack frac", r["roofline_stack"]["frac"], "nms us/img", r["nms_us_per_img"], "rows", r["detections_last_step"])
PY
}
run --height 608 --batch 32 --steps 10 --warmup 3 && run --height 416 --batch 64 --conf 0.01 --steps 10 --warmup 3 && \
run --height 320 --batch 1 --steps 50 --warmup 10 && run --height 416 --batch 1 --steps 50 --warmup 10
