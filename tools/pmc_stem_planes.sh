#!/bin/bash
# LDS / VALU / matrix-pipe counters of the stem weight-gradient kernels (tools only; counters in their own pass, kernel trace only)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/splpmc; mkdir -p $OUT; export TMPDIR=/tmp
for v in ${1:-1}; do
  cd /tmp; rm -rf $OUT/p_$v
  CUNET_NO_SIDE_STREAM=1 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU \
      --kernel-trace --output-format csv -d $OUT/p_$v -o pmc -- python $ROOT/bench.py --no-cpu-baseline --no-also --no-alone --steps 2 --warmup 1 --planner-opt stem_wgrad_planes=$v > /dev/null 2> $OUT/p_$v.err
  cd $ROOT
  python tools/pmc_summary.py "$(dirname $(ls $OUT/p_$v/*/*counter_collection.csv $OUT/p_$v/*counter_collection.csv 2>/dev/null | head -1))" "wgrad3_stem" 4
done
