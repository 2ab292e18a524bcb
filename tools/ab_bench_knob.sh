#!/bin/bash
# Whole-benchmark A/B of one environment knob on one box: A / B / A / B.
#   bash tools/ab_bench_knob.sh YOLO_AMD_CONV3_STG 3 9
cd "$GRAFT_REPO_ROOT"
K="$1"; A="$2"; B="$3"
for i in 1 2; do
  for v in "$A" "$B"; do
    export $K=$v
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-traffic --no-pipelined --steps 30 > gpurun_out/ab_knob.json 2> gpurun_out/ab_knob.err || { tail -3 gpurun_out/ab_knob.err; exit 1; }
    python - "$K=$v" <<'PY'
import json, sys
r = json.load(open("gpurun_out/ab_knob.json"))
ck = r.get("conv_kernels", {})
patch = next((v for k, v in ck.items() if "patch" in k), {})
print(sys.argv[1], "images/s", r["value"], "ms/step", r["ms_per_step"], "stages", r.get("stages_ms"), "patch us", round(patch.get("avg_launch_ms", 0) * 1e3, 2))
PY
  done
done
