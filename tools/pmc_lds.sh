#!/bin/bash
# LDS-side PMC pass over tools/conv_bench.py for the layers in $1, variant $2.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LAYERS="$1"; VAR="${2:-ns=2}"; TAG="${3:-pmcl}"
mkdir -p gpurun_out/$TAG
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/$TAG/$name -- python tools/conv_bench.py --reps 3 --layers "$LAYERS" --variants "$VAR" > gpurun_out/$TAG/$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/$TAG/$name.log; exit 1; }
}
run l1 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE && \
run l2 SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
python - "$TAG" <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(f"gpurun_out/{sys.argv[1]}/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "conv_igemm" in r["Kernel_Name"] or "conv3x3" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:36s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
PY
