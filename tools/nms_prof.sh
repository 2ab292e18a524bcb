#!/bin/bash
# per-kernel NMS times of the benchmark for the library in $YOLO_AMD_LIB (default: the tree's)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/nmsprof
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/nmsprof -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined --no-traffic --no-kernel-pass --in-flight 1 > /dev/null 2>&1
python - <<PY
import csv,glob,collections
rows=collections.defaultdict(lambda:[0,0.0])
for f in glob.glob("gpurun_out/nmsprof/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "nms" in r["Kernel_Name"]:
            k=r["Kernel_Name"].split("(")[0]; rows[k][0]+=1; rows[k][1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
print({k.replace("yolo_amd::","").replace("void ",""):round(t/n,1) for k,(n,t) in rows.items()})
PY
