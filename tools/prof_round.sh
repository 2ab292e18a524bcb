#!/bin/bash
# Round profile over the benchmark itself: kernel trace summary, then SQ counter
# passes (one rocprofv3 run per pass), aggregated per kernel family into
# gpurun_out/<tag>_mfma_counters.json and gpurun_out/<tag>_kernel_stats.txt.
#   bash tools/prof_round.sh [tag]
TAG=${1:-round3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-calibration --no-pipelined --no-traffic --no-kernel-pass --in-flight 1"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-calibration --no-pipelined --no-traffic --no-kernel-pass --in-flight 1 > $OUT/trace.json 2> $OUT/trace.err || { echo "trace pass failed"; tail -5 $OUT/trace.err; exit 1; }
run() {
  name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python bench.py $ARGS > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; exit 1; }
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 && \
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE && \
python - $OUT $TAG <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
def fam(kn):
    return kn.split("(")[0].replace("void ", "").replace("yolo_amd::", "")
# kernel stats
rows = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(f"{out}/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        rows[fam(r["Kernel_Name"])][0] += 1
        rows[fam(r["Kernel_Name"])][1] += d
tot = sum(v[1] for v in rows.values())
with open(f"gpurun_out/{tag}_kernel_stats.txt", "w") as fh:
    fh.write("rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-calibration --no-pipelined --no-traffic --no-kernel-pass --in-flight 1\n")
    fh.write(f"{'kernel':70s} {'calls':>7s} {'avg us':>9s} {'total ms':>9s} {'%':>6s}\n")
    for k, (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        fh.write(f"{k[:70]:70s} {n:7d} {t / n:9.1f} {t / 1e3:9.2f} {100 * t / tot:6.2f}\n")
print(open(f"gpurun_out/{tag}_kernel_stats.txt").read()[:3000])
# counters per kernel family, mean per launch
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "yolo_amd::" in r["Kernel_Name"]:
            acc[fam(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    m["launches"] = len(next(iter(c.values())))
    wc = m.get("SQ_WAVE_CYCLES", 0)
    if wc:
        m["wait_any_frac"] = m.get("SQ_WAIT_ANY", 0) / wc
        m["wait_inst_any_frac"] = m.get("SQ_WAIT_INST_ANY", 0) / wc
        m["active_inst_frac"] = m.get("SQ_ACTIVE_INST_ANY", 0) / wc
        m["wait_inst_lds_frac"] = m.get("SQ_WAIT_INST_LDS", 0) / wc
    # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; SQ_BUSY_CYCLES quad-cycles over SEs: report
    # matrix-pipe busy share against the kernel's clock cycles from GRBM_GUI_ACTIVE (sum over 8 XCDs)
    if m.get("GRBM_GUI_ACTIVE"):
        m["mfma_busy_per_simd_over_kernel_cycles"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (256 * 4) / (m["GRBM_GUI_ACTIVE"] / 8)
    if m.get("SQ_LDS_IDX_ACTIVE"):
        m["lds_bank_conflict_frac"] = m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"]
    res[k] = m
json.dump({"command": "rocprofv3 --pmc <pass> -- python bench.py --steps 10 --warmup 3 (batch 32, 416)",
           "note": "mean per launch; *_frac are shares of SQ_WAVE_CYCLES; GRBM_GUI_ACTIVE is summed over the 8 XCDs",
           "kernels": res}, open(f"gpurun_out/{tag}_mfma_counters.json", "w"), indent=1)
for k in sorted(res, key=lambda k: -res[k].get("SQ_WAVE_CYCLES", 0))[:6]:
    print(k, {n: round(v, 3) for n, v in res[k].items() if n.endswith("frac") or n.startswith("mfma") or n == "launches"})
PY
