#!/bin/bash
# Whole-benchmark A/B of yolo_amd/libyolo_amd_base.so (variant) against the tree's library on one box:
# base / tree / base / tree.   bash tools/ab_bench_lib.sh
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
  for v in base tree; do
    if [ $v = base ]; then export YOLO_AMD_LIB=$PWD/yolo_amd/libyolo_amd_base.so; else unset YOLO_AMD_LIB; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-traffic --no-pipelined --steps 30 > gpurun_out/ab_lib.json 2> gpurun_out/ab_lib.err || { tail -3 gpurun_out/ab_lib.err; exit 1; }
    python - "$v" <<'PY'
import json, sys
r = json.load(open("gpurun_out/ab_lib.json"))
ck = r.get("conv_kernels", {})
print(sys.argv[1], "images/s", r["value"], "ms/step", r["ms_per_step"], "stages", r.get("stages_ms"),
      {k.split("<")[-1][:14]: round(v["ms_per_step"], 3) for k, v in ck.items()})
PY
  done
done
