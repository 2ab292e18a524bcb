#!/bin/bash
# Same box: the previous revision's library, this one without the folded 1x1 layer, this one with it.
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
run() {
  timeout -k 10 200 env "$@" python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-traffic --no-pipelined 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); ck=r['conv_kernels']; print('$*'[-34:], r['value'], 'serial', r['serial']['value'], 'conv', r['stages_ms']['conv_stack'], {k[:26]:v['ms_per_step'] for k,v in ck.items() if 'stem' in k or '32, 32' in k})" || exit 1
}
for i in 1 2 3; do
run YOLO_AMD_LIB=$PWD/yolo_amd/libyolo_amd_base.so
run YOLO_AMD_STEM_TAIL=0
run YOLO_AMD_STEM_TAIL=1
done 2>&1 | tee gpurun_out/r3_stem_tail_ab.log
