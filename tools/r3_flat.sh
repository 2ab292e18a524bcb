#!/bin/bash
# flat-patch 3x3 kernel: parity cases, then A/B against the rotated-patch kernel on the benchmark's layers
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_forward.py -q -x -k "patch_kernel and (flat or private)" --timeout 300 --timeout-method thread 2>&1 | tail -5 || exit 1
V="${1:-flat=0;split=6;split=7;flat=0;split=6;split=7}"
for L in "3x3 256->512 @26" "3x3 128->256 @52" "3x3 512->1024 @13" "3x3  64->128 @104" "608 3x3"; do
  timeout -k 10 200 python tools/conv_bench.py --reps 20 --layers "$L" --variants "$V" || exit 1
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_flat.log
