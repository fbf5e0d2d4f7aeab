#!/bin/bash
# Round-3 GPU call 5: LDS-transposed epilogue of the fp32 data gradient -- node tests, config 2 timing overlapped and serial.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest "tests/test_gpu_nodes.py::test_every_node_backward_full_width" "tests/test_gpu_nodes.py::test_every_node_backward_with_quan_input" \
   "tests/test_gpu_nodes.py::test_every_node_backward_rectangular_full_width" "tests/test_gpu_nodes.py::test_every_node_backward_matches_autograd" \
   tests/test_gpu_nodes.py::test_whole_backward_composition_bench_batch -m gpu -q --maxfail=12 > $OUT/r03c5_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/r03c5_pytest.txt )
tail -6 $OUT/r03c5_pytest.txt
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-alone"
run() { local tag=$1; shift; timeout 300 "$@" > $OUT/r03c5_$tag.json 2> $OUT/r03c5_$tag.err; python -c "import json;d=json.load(open('$OUT/r03c5_$tag.json'));print('$tag', d['value'], d['ms_per_step_median'])"; }
run f32 $B
run f32_b $B
CUNET_NO_SIDE_STREAM=1 run f32_serial $B
grep -A 24 "per-class profile" $OUT/r03c5_f32_serial.err | head -30
