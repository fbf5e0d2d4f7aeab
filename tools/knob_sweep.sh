#!/bin/bash
# tools only: sweep one tuning-library environment knob around the bench line.  usage: knob_sweep.sh NAME v1 v2 ... [-- bench args]
export CUNET_LIB_PATH=$(pwd)/cu_net_amd/libcunet_hip_tuning.so
NAME=$1; shift
VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=($1); shift; done; [ "$1" == "--" ] && shift
B="python tools/bench_tuning.py --no-also --no-cpu-baseline --no-alone --steps 40 $@"
for v in "${VALS[@]}"; do
  echo -n "$NAME=$v: "
  env $NAME=$v $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step_median'])"
done
