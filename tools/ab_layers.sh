#!/bin/bash
# Layer timing A/B: yolo_amd/libyolo_amd_base.so against the tree's library, alternating, twice each.
#   bash tools/ab_layers.sh ["layer substring"]
cd "$GRAFT_REPO_ROOT"
L="${1:-3x3 }"
for i in 1 2; do
  for v in base tree; do
    if [ $v = base ]; then export YOLO_AMD_LIB=$PWD/yolo_amd/libyolo_amd_base.so; else unset YOLO_AMD_LIB; fi
    echo "== $v"
    timeout -k 10 200 python tools/conv_bench.py --variants "ns=2" --reps 30 --layers "$L" 2>&1 | grep -v "amdgpu.ids\|^layer" || exit 1
  done
done
