#!/bin/bash
# Round-3 GPU call 31: the whole GPU suite, the default bench line and the profile recipe at HEAD.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_calls_r03/r03_call12.sh
cp $OUT/r03c12_bench.json $OUT/r03c31_bench.json
bash tools/profile_r03.sh $1 > $OUT/r03c31_profile.log 2>&1
tail -3 $OUT/r03c31_profile.log | cut -c1-200
