#!/bin/bash
# First GPU pass of the round: whole GPU suite, the not-yet-collected no-alloc test, bench.
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && rm -f gpurun_out/parity_metrics.json
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_gpu_tests.log 2>&1
rc=$?; tail -15 gpurun_out/r2_gpu_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -u -m pytest tests/wip_no_alloc.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r2_noalloc.log 2>&1
rc=$?; tail -5 gpurun_out/r2_noalloc.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py > gpurun_out/r2_bench_a.json 2> gpurun_out/r2_bench_a.err
rc=$?; tail -3 gpurun_out/r2_bench_a.err; cat gpurun_out/r2_bench_a.json; exit $rc
