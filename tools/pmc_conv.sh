#!/bin/bash
# PMC passes (one rocprofv3 run per pass) over tools/conv_bench.py for the layers given in $1.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LAYERS="$1"; VAR="${2:-ns=2}"
mkdir -p gpurun_out/pmc
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc/$name -- python tools/conv_bench.py --reps 3 --layers "$LAYERS" --variants "$VAR" > gpurun_out/pmc/$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/pmc/$name.log; exit 1; }
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS && \
run tc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE && \
run sq2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU
find gpurun_out/pmc -name "*counter_collection.csv" | head
