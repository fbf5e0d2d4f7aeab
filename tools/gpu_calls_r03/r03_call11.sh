#!/bin/bash
# Round-3 GPU call 11: popcount with fat blocks; how much of the fp32 step is the statistics atomics' tail?
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_calls_r03/r03_call9.sh
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-alone --layers 16 --class-num 16 --bits-w 1"
run() { local tag=$1; shift; timeout 300 "$@" > $OUT/r03c11_$tag.json 2> $OUT/r03c11_$tag.err; python -c "import json;d=json.load(open('$OUT/r03c11_$tag.json'));print('$tag', d['value'], d['ms_per_step_median'])"; }
run popcount $B --popcount
T="python tools/bench_tuning.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-alone"
CUNET_LIB_PATH=$ROOT/cu_net_amd/libcunet_hip_tuning.so run f32_tuning $T
CUNET_LIB_PATH=$ROOT/cu_net_amd/libcunet_hip_tuning.so CUNET_CONV_DBG=1 run f32_nostats $T
CUNET_LIB_PATH=$ROOT/cu_net_amd/libcunet_hip_tuning.so CUNET_NO_SIDE_STREAM=1 run f32_tuning_serial $T
CUNET_LIB_PATH=$ROOT/cu_net_amd/libcunet_hip_tuning.so CUNET_NO_SIDE_STREAM=1 CUNET_CONV_DBG=1 run f32_nostats_serial $T
