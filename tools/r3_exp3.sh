#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
bash tools/r3_exp2.sh "var=0;var=16384;var=0;var=16384" || exit 1
YOLO_AMD_CONV3_VAR=16384 timeout -k 10 300 python -u -m pytest tests/test_gpu_forward.py -q -x -k "patch or benchmark_configurations" --timeout 300 --timeout-method thread 2>&1 | tail -5
