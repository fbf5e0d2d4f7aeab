#!/bin/bash
# Round-3 GPU call 34: same-box A/B of the working tree against HEAD's library (config 3 and config 2).
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
A3="--steps 20 --warmup 5 --no-cpu-baseline --no-also --no-alone --layers 8 --bf16-grads"
A2="--steps 40 --warmup 5 --no-cpu-baseline --no-also --no-alone"
run() { local tag=$1; shift; timeout 300 "$@" > $OUT/r03c34_$tag.json 2>/dev/null; python -c "import json;d=json.load(open('$OUT/r03c34_$tag.json'));print('$tag', d['value'], d['ms_per_step_median'])"; }
for rep in 1 2; do
  run new3_$rep python bench.py $A3
  CUNET_LIB_PATH=$ROOT/cu_net_amd/libcunet_hip_prev.so run prev3_$rep python tools/bench_tuning.py $A3
done
run new2 python bench.py $A2
CUNET_LIB_PATH=$ROOT/cu_net_amd/libcunet_hip_prev.so run prev2 python tools/bench_tuning.py $A2
rocm-smi --showclocks 2>/dev/null | head -20
