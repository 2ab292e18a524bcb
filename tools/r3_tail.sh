#!/bin/bash
# A 1x1 layer in the epilogue of the 3x3 layer in front: parity, then whole-bench A/B.
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_forward.py -x -q --timeout 120 --timeout-method thread \
  -k "epilogue_of_the_3x3 or stem_in_one_launch or fused_block or benchmark_configurations or conv_layer" 2>&1 | tail -8 || exit 1
bash tools/r3_ab_env.sh YOLO_AMD_TAIL=0 YOLO_AMD_TAIL=1 YOLO_AMD_TAIL=3
