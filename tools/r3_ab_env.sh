#!/bin/bash
# Whole-bench A/B of environment settings on one box: bash tools/r3_ab_env.sh "A=0" "YOLO_AMD_BLOCK=0" ...
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
run() {
  timeout -k 10 200 env "$@" python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-traffic --no-kernel-pass --no-pipelined 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', r['value'], 'serial', r['serial']['value'], 'conv', r['stages_ms']['conv_stack'], 'nms', r['stages_ms']['nms'])" || exit 1
}
for i in 1 2 3; do
  for v in "$@"; do run $v; done
done 2>&1 | tee gpurun_out/r3_ab_env.log
