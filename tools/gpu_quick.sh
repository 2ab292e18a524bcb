#!/bin/bash
# NMS + harness + dist tests, then the bench without the CPU baseline.
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_nms.py tests/test_gpu_harness.py tests/test_gpu_dist.py tests/test_zz_no_alloc.py -x -q --timeout 200 --timeout-method thread > gpurun_out/quick_tests.log 2>&1
rc=$?; tail -6 gpurun_out/quick_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline --no-traffic "$@" > gpurun_out/quick_bench.json 2> gpurun_out/quick_bench.err
rc=$?; tail -3 gpurun_out/quick_bench.err; python - <<'PY'
import json
r=json.load(open("gpurun_out/quick_bench.json"))
print({k:r[k] for k in ("value","ms_per_step","stages_ms","nms_us_per_img")}, "serial", r.get("serial",{}).get("value"), "pipeline class", r.get("pipelined",{}).get("value"))
for k,v in r.get("conv_kernels",{}).items(): print("  ",k,v)
PY
exit $rc
