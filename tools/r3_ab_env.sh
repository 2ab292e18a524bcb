#!/bin/bash
# Whole-bench A/B of environment settings on one box: bash tools/r3_ab_env.sh "A=0" "YOLO_AMD_BLOCK=1" ...
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
run() {
  timeout -k 10 200 env "$@" python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-traffic --no-pipelined 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); ck=r['conv_kernels']; print('$*', r['value'], 'serial', r['serial']['value'], 'conv', r['stages_ms']['conv_stack'], 'nms', r['stages_ms']['nms'], {k[:28]:v['ms_per_step'] for k,v in ck.items() if 'flat' in k or '128, 128, 32' in k or '128, 64, 32, act 1' in k})" || exit 1
}
for i in 1 2; do
  for v in "$@"; do run $v; done
done 2>&1 | tee gpurun_out/r3_ab_env.log
