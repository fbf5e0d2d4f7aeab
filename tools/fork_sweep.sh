#!/bin/bash
# tools only: cost of the fork events on the caller's stream (tuning library)
export CUNET_LIB_PATH=$(pwd)/cu_net_amd/libcunet_hip_tuning.so
B="python tools/bench_tuning.py --no-also --no-cpu-baseline --no-alone --steps 40"
for ff in 0 1; do for fa in 0 1; do
  echo -n "FORK_FLAGS=$ff FORK_AFTER=$fa: "
  CUNET_FORK_FLAGS=$ff CUNET_FORK_AFTER=$fa $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step_median'])"
done; done
