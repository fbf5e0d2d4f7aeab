#!/bin/bash
# Round-3 GPU call 9: kernel trace of the popcount forward (CU-Net-2, K = 16, bits_w = 1).
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
P="python $ROOT/bench.py --layers 2 --class-num 16 --bits-w 1 --popcount --steps 6 --warmup 3 --no-cpu-baseline --no-also --no-alone"
CUNET_NO_SIDE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/r03c9_tr -o t -- $P > /dev/null 2> $OUT/r03c9_tr.err
cd $ROOT
python tools/trace_summary.py "$(ls $OUT/r03c9_tr/*kernel_trace.csv | head -1)" 80 > $OUT/r03c9_by_grid.txt 2>&1
grep -E "ternary|conv3x3_ring|tapsplit|name" $OUT/r03c9_by_grid.txt
rm -rf $OUT/r03c9_tr
