#!/bin/bash
# whole-bench A/B of the working tree's library against yolo_amd/libyolo_amd_base.so: serial conv stack, value, the persistent kernels
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
run() {
  timeout -k 10 200 env "$@" python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-calibration --no-traffic --no-pipelined 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); ck=r['conv_kernels']; print('$*'[-30:], r['value'], 'serial', r['serial']['value'], 'conv', r['stages_ms']['conv_stack'], {k[:34]:v['ms_per_step'] for k,v in ck.items()})" || exit 1
}
for i in 1 2 3; do
run YOLO_AMD_LIB=$PWD/yolo_amd/libyolo_amd_base.so
run A=1
done 2>&1 | tee gpurun_out/r3_ab_lib.log
