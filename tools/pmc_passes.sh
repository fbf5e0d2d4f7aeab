#!/bin/bash
# PMC passes for kernel diagnosis (each --pmc set in its own run, kernel-trace only).  usage: pmc_passes.sh <tag> [env...]
set -u
TAG=$1; shift
ROOT=$(pwd)
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM" \
           "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  env "$@" timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmcx_${TAG}_$i -o pmc -- $BENCH > /dev/null 2> $ROOT/gpurun_out/pmcx_${TAG}_$i.err
  (cd $ROOT; python tools/pmc_summary.py gpurun_out/pmcx_${TAG}_$i "${PAT:-conv_kernel}" 14 > gpurun_out/pmcx_${TAG}_$i.txt 2>&1)
done
cd $ROOT
cat gpurun_out/pmcx_${TAG}_*.txt
