#!/bin/bash
# tools only (tuning library, results are WRONG with these flags): where does the data gradient's time go?
export CUNET_LIB_PATH=$(pwd)/cu_net_amd/libcunet_hip_tuning.so
export CUNET_NO_SIDE_STREAM=1
for f in 0 2 8 10 1 11 4; do
  echo -n "CUNET_CONV_DBG=$f: "
  CUNET_CONV_DBG=$f python tools/bench_tuning.py --no-also --no-cpu-baseline --no-alone --steps 10 2>&1 | grep -E "conv1x1_bwd_data |conv3x3_bwd_data |conv1x1_fwd " | awk '{printf "%s %s  ", $1, $5}'; echo
done
