#!/bin/bash
# Phase stamps of the 3x3 patch kernel on the benchmark's four map sizes (trace build of the library).
cd $GRAFT_REPO_ROOT
export YOLO_AMD_LIB=$GRAFT_REPO_ROOT/yolo_amd/libyolo_amd_trace.so
for L in "3x3  64->128 @104" "3x3 128->256 @52" "3x3 256->512 @26" "3x3 512->1024 @13"; do
  timeout -k 10 100 python tools/conv3_trace.py --layer "$L" --batch 32 2>&1 | grep -v "amdgpu.ids\|Warning\|ret = \|f\"{" || exit 1
done
