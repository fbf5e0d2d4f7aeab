#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run from the repo root through gpurun):
#   1. --kernel-trace --stats of the bench command  -> gpurun_out/prof_<tag>/
#   usage: profile_round.sh <tag> <commit>
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (own runs, kernel-trace only) -> gpurun_out/pmc_<tag>_{rd,wr}/
# and print the per-(kernel, grid) summaries.  Copy what should be judged into profiles/.
set -u
TAG=${1:-r01}
ROOT=$(pwd)
export TMPDIR=/tmp
COMMIT=${2:-unknown}
BENCH="python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-alone"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$TAG -o t -- $BENCH > $ROOT/gpurun_out/prof_$TAG.json 2> $ROOT/gpurun_out/prof_$TAG.err
# the same with the weight-gradient side stream off: every kernel ALONE on the GPU (the per-kernel durations behind roofline.alone)
CUNET_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${TAG}_serial -o t -- $BENCH > $ROOT/gpurun_out/prof_${TAG}_serial.json 2> $ROOT/gpurun_out/prof_${TAG}_serial.err
BENCH2="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-also --no-alone"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}_rd -o pmc -- $BENCH2 > /dev/null 2> $ROOT/gpurun_out/pmc_${TAG}_rd.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}_wr -o pmc -- $BENCH2 > /dev/null 2> $ROOT/gpurun_out/pmc_${TAG}_wr.err
cd $ROOT
tail -1 gpurun_out/prof_$TAG.json
python tools/trace_summary.py $(ls gpurun_out/prof_$TAG/*kernel_trace.csv | head -1) 60 > gpurun_out/prof_${TAG}_by_grid.txt
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_rd '' 40 > gpurun_out/pmc_${TAG}_rd.txt 2>&1
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_wr '' 40 > gpurun_out/pmc_${TAG}_wr.txt 2>&1
python tools/trace_summary.py $(ls gpurun_out/prof_${TAG}_serial/*kernel_trace.csv | head -1) 60 > gpurun_out/prof_${TAG}_serial_by_grid.txt
head -30 gpurun_out/prof_${TAG}_by_grid.txt
python tools/pmc_traffic.py gpurun_out/pmc_${TAG}_rd gpurun_out/pmc_${TAG}_wr gpurun_out/traffic_$TAG.json 2,68,24,f32 $COMMIT > gpurun_out/traffic_$TAG.txt 2>&1
cat gpurun_out/traffic_$TAG.txt
