#!/bin/bash
# Thin 3x3 kernel: parity, then A/B of the working tree's library against yolo_amd/libyolo_amd_base.so (stand-alone layer).
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_forward.py -x -q --timeout 120 --timeout-method thread \
  -k "thin or fused_block or first_residual or benchmark_configurations or stem_in_one" 2>&1 | tail -3 || exit 1
for i in 1 2; do
  YOLO_AMD_LIB=$PWD/yolo_amd/libyolo_amd_base.so timeout -k 10 120 python tools/conv_bench.py --reps 30 --layers "3x3  32->64" --variants "ns=2" 2>&1 | grep "32->64" | sed 's/^/base /' || exit 1
  timeout -k 10 120 python tools/conv_bench.py --reps 30 --layers "3x3  32->64" --variants "ns=2" 2>&1 | grep "32->64" | sed 's/^/new  /' || exit 1
done | tee gpurun_out/r3_thin.log
