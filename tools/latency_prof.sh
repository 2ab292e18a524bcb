#!/bin/bash
# Kernel statistics of the batch-1 video configuration.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/lat && mkdir -p $R/gpurun_out/lat
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/lat -o lat --output-format csv -- python3 $R/tools/latency_run.py 320 > $R/gpurun_out/lat/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/lat/**/lat_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per frame %.1f us" % (tot / 50 / 1e3))
for r in rows[:14]:
    print("%-90s calls %5s avg %8.1f us  %5.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
