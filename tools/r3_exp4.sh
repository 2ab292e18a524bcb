#!/bin/bash
# correctness of the persistent kernels after a change, then the bench line with traffic
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_forward.py tests/test_gpu_train_bn.py tests/test_gpu_harness.py tests/test_zz_no_alloc.py -q -x --timeout 300 --timeout-method thread 2>&1 | tail -4 || exit 1
timeout -k 10 600 python bench.py --no-cpu-baseline --layer-table gpurun_out/r3_layer_roofline_b.txt > gpurun_out/r3_bench_b.json 2> gpurun_out/r3_bench_b.err || { tail -5 gpurun_out/r3_bench_b.err; exit 1; }
python - <<'PY'
import json
r=json.load(open("gpurun_out/r3_bench_b.json"))
print({k:r[k] for k in ("value","ms_per_step","stages_ms","nms_us_per_img")}, "serial", r["serial"]["value"])
for k in ("conv_stem_kernel","conv3x3_thin_kernel"):
    print(k, r["conv_kernels"][k], {a:round(b/1e6,1) for a,b in r["hbm_traffic_per_launch"][k].items()})
PY
