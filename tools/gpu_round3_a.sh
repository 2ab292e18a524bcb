#!/bin/bash
# Whole GPU suite, smoke, then the full bench line with the per-launch roofline table.
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && rm -f gpurun_out/parity_metrics.json
timeout -k 10 1000 python -u -m pytest tests -m gpu -q --timeout 400 --timeout-method thread --durations=8 > gpurun_out/r3_gpu_tests.log 2>&1
rc=$?; tail -15 gpurun_out/r3_gpu_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || exit 1
timeout -k 10 600 python bench.py --layer-table gpurun_out/r3_layer_roofline.txt > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
rc=$?; tail -3 gpurun_out/r3_bench.err; [ $rc -ne 0 ] && exit $rc
python - <<'PY'
import json
r=json.load(open("gpurun_out/r3_bench.json"))
print({k:r[k] for k in ("value","ms_per_step","stages_ms","nms_us_per_img")}, "serial", r["serial"]["value"], r["roofline"]["frac"], r["roofline_stack"]["frac"], r["roofline"]["traffic"], r["roofline_composite"], r["cpu_baseline"])
PY
