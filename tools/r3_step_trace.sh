#!/bin/bash
# the ordered list of GPU kernels of one step (between two launches of the stem kernel)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/step_trace; rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pipelined --no-traffic --no-kernel-pass --in-flight ${1:-1} > $OUT/run.json 2> $OUT/run.err || { tail -5 $OUT/run.err; exit 1; }
python - $OUT <<'PY'
import csv, glob, sys
rows=[]
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "conv_stem_kernel" in r["Kernel_Name"]]
a,b=idx[-2],idx[-1]
t0=int(rows[a]["Start_Timestamp"])
prev_end=None
for r in rows[a:b]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    name=r["Kernel_Name"].replace("void ","").replace("yolo_amd::","")[:60]
    gap=(s-prev_end)/1e3 if prev_end else 0
    print(f"{(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:7.1f}  gap {gap:6.1f}  q{r.get('Queue_Id','?')} {name}")
    prev_end=e
print("step:", (int(rows[b]["Start_Timestamp"])-t0)/1e3, "us")
PY
