#!/bin/bash
# HBM-side traffic of the conv stack: two rocprofv3 --pmc passes (reads, writes) over the
# benchmark itself, summed over the conv kernels and divided by the steps run.
# Writes gpurun_out/conv_traffic.json (copy it to profiles/ to have bench.py report it).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
STEPS=5; WARM=2
mkdir -p gpurun_out/pmcb
run() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmcb/$name -- python bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-pipelined > gpurun_out/pmcb/$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/pmcb/$name.log; exit 1; }
}
run rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum && \
run wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum && \
python - $STEPS $WARM <<'PY'
import csv, glob, json, sys, collections
steps = int(sys.argv[1]) + int(sys.argv[2])
tot = collections.defaultdict(float)
n_launch = 0
for f in glob.glob("gpurun_out/pmcb/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "yolo_amd::conv" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "TCC_EA0_RDREQ_sum":
                n_launch += 1
rd, rd32 = tot["TCC_EA0_RDREQ_sum"], tot["TCC_EA0_RDREQ_32B_sum"]
wr, wr64 = tot["TCC_EA0_WRREQ_sum"], tot["TCC_EA0_WRREQ_64B_sum"]
# reads: 16-byte-per-lane streaming loads issue 128-byte requests that the counter tallies
# as one (64 B in FETCH_SIZE): count 128 B per non-32B request; writes: 64 B / 32 B exact
read_b = (rd - rd32) * 128 + rd32 * 32
write_b = wr64 * 64 + (wr - wr64) * 32
out = {"steps": steps, "conv_launches": n_launch, "read_bytes_per_step": read_b / steps,
       "write_bytes_per_step": write_b / steps,
       "hbm_bytes_per_launch": (read_b + write_b) / max(n_launch, 1),
       "counters": dict(tot),
       "method": "rocprofv3 --pmc, separate passes for TCC_EA0_RDREQ(+_32B) and TCC_EA0_WRREQ(+_64B) over bench.py; "
                 "read bytes = 128 B per non-32B request (gfx950 tallies the 128-byte requests of 16-byte-per-lane "
                 "loads once), write bytes = 64 B / 32 B per request; Infinity Cache hits are included"}
json.dump(out, open("gpurun_out/conv_traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "counters" and k != "method"}))
PY
