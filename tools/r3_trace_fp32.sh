#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export YOLO_AMD_LIB=$PWD/yolo_amd/libyolo_amd_trace.so
for L in "1x1 256->255 @52 fp32" "1x1 512->255 @26 fp32" "1x1 256->128 @52" "1x1 1024->512 @13"; do
timeout -k 10 120 python tools/conv3_trace.py --layer "$L" 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r3_trace_fp32.log
