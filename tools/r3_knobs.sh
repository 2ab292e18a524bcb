#!/bin/bash
# Whole-bench A/B of dispatch knobs with three batches in flight (value) and one after the other (serial).
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
run() {
  timeout -k 10 200 env "$@" python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-traffic --no-kernel-pass --no-pipelined 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', r['value'], 'serial', r['serial']['value'], 'conv', r['stages_ms']['conv_stack'], 'nms', r['stages_ms']['nms'])" || exit 1
}
for i in 1 2; do
run A=0
run YOLO_AMD_CONV_BN128_MIN=0
run YOLO_AMD_CONV_BN128_MIN=256
run YOLO_AMD_CONV_BK=32
run YOLO_AMD_CONV_NSTAGE=3
done 2>&1 | tee gpurun_out/r3_knobs.log
