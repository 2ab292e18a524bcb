#!/bin/bash
# Phase stamps of the im2col kernel on the stride-2 3x3 layers and two 1x1 layers at batch 32 (trace build).
cd $GRAFT_REPO_ROOT
export YOLO_AMD_LIB=$GRAFT_REPO_ROOT/yolo_amd/libyolo_amd_trace.so
for L in "3x3s2 64->128 @208" "3x3s2 128->256 @104" "3x3s2 256->512 @52" "3x3s2 512->1024 @26" "1x1 256->128 @52" "1x1 512->256 @26"; do
  timeout -k 10 100 python tools/conv3_trace.py --layer "$L" --batch 32 2>&1 | grep -E "batch 32:|all workgroups|epilogue:|first 512|the rest|successor" || exit 1
done
