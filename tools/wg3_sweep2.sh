#!/bin/bash
# Tuning sweep of the 3x3 ring weight gradient's planner knobs (tuning library; never a product number).
export CUNET_LIB_PATH=$(pwd)/cu_net_amd/libcunet_hip_tuning.so
run() {
  local tag=$1; shift
  env "$@" python tools/bench_tuning.py --no-also --no-cpu-baseline --steps 30 --warmup 4 2> gpurun_out/sweep2_$tag.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$tag', d['value'], d['ms_per_step'])"
  grep -E "conv3x3_bwd_weight|misc " gpurun_out/sweep2_$tag.err | head -2
}
mkdir -p gpurun_out
run off CUNET_WG3_3X3=0
for rows in 6 12 24; do for mw in 2 16 32; do
  run "r${rows}_w${mw}" CUNET_WG3_3X3_ROWS=$rows CUNET_WG3_3X3_MIN_W=$mw
done; done
