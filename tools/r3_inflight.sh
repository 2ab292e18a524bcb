#!/bin/bash
# Throughput against the number of batches in flight, one box.
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for i in 1 2; do
for F in 2 3 4 5 6; do
  timeout -k 10 200 python bench.py --steps 60 --warmup 12 --in-flight $F --no-cpu-baseline --no-traffic --no-pipelined --no-kernel-pass 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in flight', $F, r['value'], r['ms_per_step'])" || exit 1
done
done 2>&1 | tee gpurun_out/r3_inflight.log
