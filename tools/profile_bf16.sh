#!/bin/bash
# rocprofv3 kernel stats of the CU-Net-8 bf16-storage step (BASELINE config 3).  usage: profile_bf16.sh <tag>
set -u
TAG=${1:-r02}
ROOT=$(pwd)
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --layers 8 --bf16-grads --steps 6 --warmup 3 --no-cpu-baseline --no-also --no-alone"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${TAG}_bf16 -o t -- $BENCH > $ROOT/gpurun_out/prof_${TAG}_bf16.json 2> $ROOT/gpurun_out/prof_${TAG}_bf16.err
cd $ROOT
python tools/trace_summary.py $(ls gpurun_out/prof_${TAG}_bf16/*kernel_trace.csv | head -1) 60 > gpurun_out/prof_${TAG}_bf16_by_grid.txt
head -25 gpurun_out/prof_${TAG}_bf16_by_grid.txt
tail -1 gpurun_out/prof_${TAG}_bf16.json | cut -c1-400
