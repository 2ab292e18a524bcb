#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export YOLO_AMD_LIB=$PWD/yolo_amd/libyolo_amd_trace.so YOLO_AMD_CONV3_FLAT=1
for L in "3x3 256->512 @26" "3x3 128->256 @52" "3x3 512->1024 @13" "3x3  64->128 @104"; do
  timeout -k 10 120 python tools/tap_trace.py --layer "$L" || exit 1
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_tap_trace.log
