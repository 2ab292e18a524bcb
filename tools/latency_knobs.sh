#!/bin/bash
# Small-batch latency under the im2col kernel's development knobs, one process per setting.
cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; env $1 timeout -k 10 160 python tools/latency_time.py 320x1 416x1 416x2 416x4 416x8 416x16 2>&1 | grep -v amdgpu.ids | tail -1; }
run "YOLO_AMD_CONV_BN64_MIN=0" && run "YOLO_AMD_CONV_BN64_MIN=64" && run "YOLO_AMD_CONV_BN64_MIN=128" && run "YOLO_AMD_CONV_BN64_MIN=256" \
 && run "YOLO_AMD_CONV_BN64_MIN=512" && run "YOLO_AMD_CONV_BN64_MIN=100000" \
 && run "YOLO_AMD_SPLITK_MIN_KT=32" && run "YOLO_AMD_SPLITK_MIN_KT=16" && run "YOLO_AMD_SPLITK_MIN_KT=32 YOLO_AMD_SPLITK_MAX_WGS=256" \
 && run "YOLO_AMD_SPLITK_FILL=1024" && run "YOLO_AMD_SPLITK_FILL=256"
