#!/bin/bash
# Whole GPU suite, smoke, the full bench line, then the round profile.
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && rm -f gpurun_out/parity_metrics.json
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_gpu_tests.log 2>&1
rc=$?; tail -4 gpurun_out/r2_gpu_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || exit 1
timeout -k 10 600 python bench.py > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err
rc=$?; tail -3 gpurun_out/r2_bench.err; [ $rc -ne 0 ] && exit $rc
python - <<'PY'
import json
r=json.load(open("gpurun_out/r2_bench.json"))
print({k:r[k] for k in ("value","ms_per_step","stages_ms","nms_us_per_img")}, "serial", r["serial"]["value"], r["roofline"]["frac"], r["roofline_stack"]["frac"], r["roofline"]["traffic"], "pipeline class", r.get("pipelined",{}).get("value"))
PY
bash tools/prof_round2.sh round2 > gpurun_out/r2_prof.log 2>&1; tail -12 gpurun_out/r2_prof.log
