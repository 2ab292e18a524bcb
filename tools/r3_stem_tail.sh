#!/bin/bash
# The stem with the 1x1 layer folded in: parity, then whole-bench A/B against the separate launch.
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_forward.py -x -q --timeout 120 --timeout-method thread \
  -k "stem_in_one_launch or first_layer or fused_block or benchmark_configurations" 2>&1 | tail -5 || exit 1
bash tools/r3_ab_env.sh YOLO_AMD_STEM_TAIL=0 YOLO_AMD_STEM_TAIL=1
