#!/bin/bash
# Round-3 GPU call 32: (a) the fp32 data gradients with the LDS-tile epilogue at 16 instead of 12 waves per CU (probe build against
# the tuning build, both -DCUNET_TUNING); (b) hand-over group size of the weight gradients after the round's launch-count changes.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
A="--steps 40 --warmup 5 --no-cpu-baseline --no-also --no-alone"
run() { local tag=$1; shift; timeout 300 "$@" > $OUT/r03c32_$tag.json 2>/dev/null; python -c "import json;d=json.load(open('$OUT/r03c32_$tag.json'));print('$tag', d['value'], d['ms_per_step_median'])"; }
for rep in 1 2; do
  CUNET_LIB_PATH=$ROOT/cu_net_amd/libcunet_hip_probe.so run w16_$rep python tools/bench_tuning.py $A
  CUNET_LIB_PATH=$ROOT/cu_net_amd/libcunet_hip_tuning.so run w12_$rep python tools/bench_tuning.py $A
done
CUNET_LIB_PATH=$ROOT/cu_net_amd/libcunet_hip_probe.so run w16_q python tools/bench_tuning.py $A --layers 16 --class-num 16 --bits-w 1 --steps 10
CUNET_LIB_PATH=$ROOT/cu_net_amd/libcunet_hip_tuning.so run w12_q python tools/bench_tuning.py $A --layers 16 --class-num 16 --bits-w 1 --steps 10
for g in 2 4 8; do
  run f32_g$g python bench.py $A --planner-opt wgrad_fork_group=$g
  run bf16_g$g python bench.py $A --layers 8 --bf16-grads --steps 20 --planner-opt wgrad_fork_group=$g
done
