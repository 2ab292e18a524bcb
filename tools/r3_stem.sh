#!/bin/bash
# stem kernel after a change: its parity tests, then the bench's per-kernel times (serial, no traffic passes)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_forward.py -q -x -k "stem or first or benchmark_configurations or fused_block" --timeout 300 --timeout-method thread 2>&1 | tail -4 || exit 1
timeout -k 10 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-traffic > gpurun_out/r3_stem_bench.json 2>/dev/null || exit 1
python - <<'PY'
import json
r=json.load(open("gpurun_out/r3_stem_bench.json"))
print({k:r[k] for k in ("value","ms_per_step","stages_ms")}, "serial", r["serial"]["value"])
for k,v in r["conv_kernels"].items(): print(k, v)
PY
