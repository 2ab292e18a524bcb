#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
V="${1:-fvar=0;fvar=1;fvar=0;fvar=1}"
for L in "3x3 256->512 @26" "3x3 128->256 @52" "3x3 512->1024 @13" "3x3  64->128 @104"; do
  timeout -k 10 200 python tools/conv_bench.py --reps 20 --layers "$L" --variants "$V" || exit 1
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_fvar.log
