#!/bin/bash
# Round-3 GPU call 14: bf16 forward epilogue through LDS tiles; two-rank config-4 test.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_dp.py tests/test_gpu_parity.py::test_bf16_inference_forward_close_to_fp32 tests/test_gpu_parity.py::test_bf16_activation_train_step_tracks_fp32 \
   tests/test_gpu_parity.py::test_bf16_3x3_row_ring_agrees_with_the_other_bf16_kernels "tests/test_gpu_nodes.py::test_every_node_backward_full_width_bf16_activations" "tests/test_gpu_configs.py::test_config3_cu_net8_k68" -m gpu -q --maxfail=12 > $OUT/r03c14_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/r03c14_pytest.txt )
tail -8 $OUT/r03c14_pytest.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-alone"
run() { local tag=$1; shift; timeout 300 "$@" > $OUT/r03c14_$tag.json 2> $OUT/r03c14_$tag.err; python -c "import json;d=json.load(open('$OUT/r03c14_$tag.json'));print('$tag', d['value'], d['ms_per_step_median'])"; }
run bf16 $B --layers 8 --bf16-grads
run bf16_fwd $B --forward-only --bf16
grep -E "conv1x1_fwd_bf16|conv3x3_fwd_bf16" $OUT/r03c14_bf16.err
