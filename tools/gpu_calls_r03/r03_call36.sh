#!/bin/bash
# Round-3 GPU call 36: fused-MSE epilogue in its own instantiation (no spills in the 128-column fp32 forward kernel) -- same-box A/B
# against HEAD's library first, then the whole GPU suite.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
A2="--steps 40 --warmup 5 --no-cpu-baseline --no-also --no-alone"
run() { local tag=$1; shift; timeout 200 "$@" > $OUT/r03c36_$tag.json 2>/dev/null; python -c "import json;d=json.load(open('$OUT/r03c36_$tag.json'));print('$tag', d['value'], d['ms_per_step_median'])"; }
run new2 python bench.py $A2
CUNET_LIB_PATH=$ROOT/cu_net_amd/libcunet_hip_prev.so run prev2 python tools/bench_tuning.py $A2
run new2b python bench.py $A2
run newfwd python bench.py $A2 --forward-only
CUNET_LIB_PATH=$ROOT/cu_net_amd/libcunet_hip_prev.so run prevfwd python tools/bench_tuning.py $A2 --forward-only
( timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 > $OUT/r03c36_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/r03c36_pytest.txt )
tail -4 $OUT/r03c36_pytest.txt
