#!/bin/bash
# Round-3 GPU call 6: does a wgrad3 that leaves register room for a co-resident data-gradient wave help the overlapped step?
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
T="python tools/bench_tuning.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-alone"
run() { local tag=$1; shift; timeout 300 "$@" > $OUT/r03c6_$tag.json 2> $OUT/r03c6_$tag.err; python -c "import json;d=json.load(open('$OUT/r03c6_$tag.json'));print('$tag', d['value'], d['ms_per_step_median'], d['library_path'])"; }
CUNET_LIB_PATH=$ROOT/cu_net_amd/libcunet_hip_tuning.so run tuning $T
CUNET_LIB_PATH=$ROOT/cu_net_amd/libcunet_hip_probe.so run probe_lb3 $T
CUNET_LIB_PATH=$ROOT/cu_net_amd/libcunet_hip_probe.so CUNET_NO_SIDE_STREAM=1 run probe_lb3_serial $T
CUNET_LIB_PATH=$ROOT/cu_net_amd/libcunet_hip_tuning.so CUNET_SIDE_PRIO=0 run tuning_prio0 $T
CUNET_LIB_PATH=$ROOT/cu_net_amd/libcunet_hip_tuning.so CUNET_SIDE_PRIO=2 run tuning_prio2 $T
