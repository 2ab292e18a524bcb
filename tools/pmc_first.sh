#!/bin/bash
# PMC passes (one rocprofv3 run per pass) over tools/first_bench.py.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmcf
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmcf/$name -- python tools/first_bench.py > gpurun_out/pmcf/$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/pmcf/$name.log; exit 1; }
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS && \
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES && \
run tc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_STALL_sum GRBM_GUI_ACTIVE TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmcf/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "conv_first" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:32s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
PY
