#!/bin/bash
# Patch-kernel timing variants and ablations on the four 13 * 2^n maps.
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for L in "3x3 256->512 @26" "3x3 128->256 @52" "3x3 512->1024 @13" "3x3  64->128 @104"; do
  timeout -k 10 200 python tools/conv_bench.py --reps 20 --layers "$L" --variants "var=0;var=33;var=65;var=129;var=2;var=4;var=67;var=0" || exit 1
  timeout -k 10 200 python tools/conv_bench.py --reps 20 --layers "$L" --variants "var=0;var=256;var=512;var=768;var=1792;var=2048;var=2304" || exit 1
done 2>&1 | tee gpurun_out/r3_exp1.log
