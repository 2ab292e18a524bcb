#!/bin/bash
# Per-dispatch durations (rocprofv3 kernel trace) of the 104 x 104 layers with and without the folded 1x1 layer.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for DEV in 0; do


OUT=gpurun_out/tail_trace; rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined --no-traffic --no-kernel-pass --in-flight 1 > $OUT/trace.json 2> $OUT/trace.err || { echo "trace pass failed"; tail -5 $OUT/trace.err; exit 1; }
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(f"{out}/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("yolo_amd::", "")
        if "flat" in kn and "1048576" in str(r.get("Grid_Size") or r.get("Grid_Size_X")):
            grid = r.get("Grid_Size") or r.get("Grid_Size_X")
            acc[(kn, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open("gpurun_out/r3_tail_trace.txt", "w") as fh:
    for (kn, grid), v in sorted(acc.items()):
        v = v[len(v) // 4:]
        fh.write(f"{kn[:64]:64s} grid {grid:>9s} calls {len(v):4d} avg {sum(v) / len(v):7.1f} us min {min(v):7.1f}\n")
print(open("gpurun_out/r3_tail_trace.txt").read())
PY
done
