#!/bin/bash
# Dynamic instruction mix per kernel (tools only; counters in their own pass, kernel trace only, side stream off so that every kernel is alone):
#   pmc_inst_mix.sh <name filter> [bench flags]
FILT=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/imix; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp; rm -rf $OUT/p
CUNET_NO_SIDE_STREAM=1 timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES \
    --kernel-trace --output-format csv -d $OUT/p -o pmc -- python $ROOT/bench.py --no-cpu-baseline --no-also --no-alone --steps 2 --warmup 1 "$@" > /dev/null 2> $OUT/p.err
cd $ROOT
python tools/pmc_summary.py "$(dirname $(ls $OUT/p/*/*counter_collection.csv $OUT/p/*counter_collection.csv 2>/dev/null | head -1))" "$FILT" 12
