#!/bin/bash
# Round-3 GPU call 4: LDS-transposed epilogue of the bf16 data gradient -- node tests, config 3 timing.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 600 python -m pytest "tests/test_gpu_nodes.py::test_every_node_backward_full_width_bf16_gradient_tensors" "tests/test_gpu_nodes.py::test_every_node_backward_rectangular_full_width" \
   "tests/test_gpu_nodes.py::test_every_node_backward_full_width_wgrad3" tests/test_gpu_exact.py::test_fused_head_loss_equals_the_separate_loss_pass -m gpu -q --maxfail=12 > $OUT/r03c4_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/r03c4_pytest.txt )
tail -6 $OUT/r03c4_pytest.txt
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-alone"
run() { local tag=$1; shift; timeout 300 "$@" > $OUT/r03c4_$tag.json 2>/dev/null; python -c "import json;d=json.load(open('$OUT/r03c4_$tag.json'));print('$tag', d['value'], d['ms_per_step_median'])"; }
run bf16 $B --layers 8 --bf16-grads --steps 20
run bf16_b $B --layers 8 --bf16-grads --steps 20
run bf16_l2 $B --bf16-grads
