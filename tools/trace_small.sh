#!/bin/bash
# Phase stamps of the im2col kernel on batch-1 layers (trace build of the library).
cd $GRAFT_REPO_ROOT
export YOLO_AMD_LIB=$GRAFT_REPO_ROOT/yolo_amd/libyolo_amd_trace.so
export YOLO_AMD_CONV3=0
for K in "A=0" "YOLO_AMD_CONV_NSTAGE=4" "YOLO_AMD_CONV_NSTAGE=4 YOLO_AMD_CONV_BK=64" "YOLO_AMD_CONV_NSTAGE=3 YOLO_AMD_CONV_BK=64" "YOLO_AMD_CONV_AMUL=1" "YOLO_AMD_CONV_BN=32"; do
for L in "3x3 128->256 @52" "1x1 512->256 @26"; do
  echo "== $K"
  env $K timeout -k 10 100 python tools/conv3_trace.py --layer "$L" --batch 1 2>&1 | grep -E "batch 1:|all workgroups" || exit 1
done; done
