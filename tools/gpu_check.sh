#!/bin/bash
# Run GPU test files one after another on the GPU box, each under its own time
# limit.  A plain test failure (exit 1) lets the next file run; a timeout, abort
# or fault ends the script there.
mkdir -p gpurun_out
for f in "$@"; do
  name=$(basename "$f" .py)
  echo "=== $f"
  timeout -k 10 420 python -u -m pytest "$f" -x -v --timeout 150 --timeout-method thread \
      > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "rc=$rc"
  tail -n 25 "gpurun_out/$name.log"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then
    echo "stopping: $f ended with $rc"
    exit $rc
  fi
done
exit 0
