#!/bin/bash
# A/B of the working tree's library against yolo_amd/libyolo_amd_base.so (tools/build_base.sh)
# on single layers, same box, base / new / base / new.
#   bash tools/ab_conv.sh "<layer filter>" "<variants>" [reps]
cd "$GRAFT_REPO_ROOT"
L="$1"; V="${2:-ns=2}"; R="${3:-30}"
for i in 1 2; do
  echo "--- base"; YOLO_AMD_LIB=$PWD/yolo_amd/libyolo_amd_base.so timeout -k 10 300 python tools/conv_bench.py --reps $R --layers "$L" --variants "$V" || exit 1
  echo "--- new";  timeout -k 10 300 python tools/conv_bench.py --reps $R --layers "$L" --variants "$V" || exit 1
done
