#!/bin/bash
# Tuning sweep of the wgrad3 planner knobs on the GPU box (uses the -DCUNET_TUNING library; never a product number).
#   gpurun -- 'bash tools/wg3_sweep.sh "<extra bench flags>"'
export CUNET_LIB_PATH=$(pwd)/cu_net_amd/libcunet_hip_tuning.so
EXTRA=${1:-}
run() {
  local tag=$1; shift
  env "$@" python tools/bench_tuning.py --no-also --no-cpu-baseline --steps 30 --warmup 4 $EXTRA 2> gpurun_out/sweep_$tag.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline'] or {}
print('$tag', d['value'], d['ms_per_step'], r.get('kernel'), r.get('avg_launch_us'), r.get('frac'))"
}
mkdir -p gpurun_out
run wg2 CUNET_WG3=0
for mm in 0 32768; do for mc in 2 4 8; do for sm in 128 256; do
  run "wg3_m${mm}_c${mc}_s${sm}" CUNET_WG3_MIN_M=$mm CUNET_WG3_MIN_CHUNKS=$mc CUNET_WG3_SMAX=$sm
done; done; done
