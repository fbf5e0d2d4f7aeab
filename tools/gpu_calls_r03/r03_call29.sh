#!/bin/bash
# Round-3 GPU call 29: fp32 1x1 data gradient (x requested in front of the tile's MFMAs, three-pass LDS epilogue) -- fp32 tests, then
# same-box A/B against the previous library (cu_net_amd/libcunet_hip_prev.so).
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest "tests/test_gpu_nodes.py::test_every_node_backward_full_width_wgrad3" "tests/test_gpu_nodes.py::test_every_node_backward_rectangular_full_width" \
    "tests/test_gpu_nodes.py::test_whole_backward_composition_bench_batch" tests/test_gpu_quant.py "tests/test_gpu_parity.py::test_full_width_matches_reference" \
    -m gpu -q --maxfail=12 > $OUT/r03c29_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/r03c29_pytest.txt )
tail -4 $OUT/r03c29_pytest.txt
A="--steps 40 --warmup 5 --no-cpu-baseline --no-also --no-alone"
run() { local tag=$1; shift; timeout 300 "$@" > $OUT/r03c29_$tag.json 2>/dev/null; python -c "import json;d=json.load(open('$OUT/r03c29_$tag.json'));print('$tag', d['value'], d['ms_per_step_median'])"; }
for rep in 1 2; do
  run new_$rep python bench.py $A
  CUNET_LIB_PATH=$ROOT/cu_net_amd/libcunet_hip_prev.so run prev_$rep python tools/bench_tuning.py $A
done
run new_q python bench.py $A --layers 16 --class-num 16 --bits-w 1 --steps 10
CUNET_LIB_PATH=$ROOT/cu_net_amd/libcunet_hip_prev.so run prev_q python tools/bench_tuning.py $A --layers 16 --class-num 16 --bits-w 1 --steps 10
