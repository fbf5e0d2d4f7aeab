#!/bin/bash
# Round-3 GPU call 13: bf16 row ring + bf16 heads -- tests and config 3 timing.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py::test_bf16_3x3_row_ring_agrees_with_the_other_bf16_kernels "tests/test_gpu_nodes.py::test_every_node_backward_full_width_bf16_gradient_tensors" \
   "tests/test_gpu_nodes.py::test_every_node_backward_full_width_wgrad3" tests/test_gpu_exact.py::test_fused_head_loss_equals_the_separate_loss_pass "tests/test_gpu_configs.py::test_config3_cu_net8_k68" \
   "tests/test_gpu_configs.py::test_config4_cu_net8_k16_rank_shard" tests/test_gpu_parity.py::test_bf16_inference_forward_close_to_fp32 tests/test_gpu_parity.py::test_bf16_activation_train_step_tracks_fp32 -m gpu -q --maxfail=12 > $OUT/r03c13_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/r03c13_pytest.txt )
tail -12 $OUT/r03c13_pytest.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-alone --layers 8 --bf16-grads"
run() { local tag=$1; shift; timeout 300 "$@" > $OUT/r03c13_$tag.json 2> $OUT/r03c13_$tag.err; python -c "import json;d=json.load(open('$OUT/r03c13_$tag.json'));print('$tag', d['value'], d['ms_per_step_median'])"; }
run bf16 $B
T="python tools/bench_tuning.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-alone --layers 8 --bf16-grads"
CUNET_B16_RING=0 run bf16_noring $T
run bf16_fwd python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-alone --forward-only --bf16
grep -E "conv3x3_fwd_bf16|conv1x1_bwd_data_bf16|conv1x1_bwd_data " $OUT/r03c13_bf16.err $OUT/r03c13_bf16_noring.err
