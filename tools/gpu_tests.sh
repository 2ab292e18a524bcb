#!/bin/bash
# the whole GPU suite (no -x: all failures at once), then smoke
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && rm -f gpurun_out/parity_metrics.json
timeout -k 10 1000 python -u -m pytest tests -m gpu -q --timeout 400 --timeout-method thread > gpurun_out/r3_gpu_tests.log 2>&1
rc=$?; grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r3_gpu_tests.log | tail -15; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
