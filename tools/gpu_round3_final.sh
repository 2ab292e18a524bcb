#!/bin/bash
# End-of-round evidence: whole GPU suite + smoke, the bench line with traffic and the per-launch table, the
# rocprofv3 kernel trace and counter passes.  Copies to keep go to profiles/ by hand.
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && rm -f gpurun_out/parity_metrics.json
timeout -k 10 1000 python -u -m pytest tests -m gpu -q --timeout 400 --timeout-method thread > gpurun_out/r3_gpu_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r3_gpu_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 || exit 1
timeout -k 10 600 python bench.py --layer-table gpurun_out/round3_layer_roofline.txt > gpurun_out/round3_bench.json 2> gpurun_out/round3_bench.err
rc=$?; tail -2 gpurun_out/round3_bench.err; [ $rc -ne 0 ] && exit $rc
python - <<'PY'
import json
r=json.load(open("gpurun_out/round3_bench.json"))
print({k:r[k] for k in ("value","ms_per_step","stages_ms","nms_us_per_img")}, "serial", r["serial"]["value"], r["roofline"]["frac"], r["roofline"]["traffic"], r["roofline_stack"]["frac"])
PY
bash tools/prof_round.sh round3 2>&1 | tail -25
