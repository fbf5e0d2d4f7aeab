#!/bin/bash
# Round-3 GPU call 21: chunk ring in the bf16 forward kernel (+ whole-tile look-ahead in the 1x1 data gradient) -- bf16 tests, then
# same-box A/B against the previous commit's library (cu_net_amd/libcunet_hip_prev.so).
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest "tests/test_gpu_nodes.py::test_every_node_backward_full_width_bf16_gradient_tensors" "tests/test_gpu_nodes.py::test_every_node_backward_full_width_bf16_activations" \
    "tests/test_gpu_configs.py::test_config3_cu_net8_k68" "tests/test_gpu_configs.py::test_config4_cu_net8_k16_rank_shard" tests/test_gpu_exact.py \
    tests/test_gpu_parity.py -k "bf16 or pair or heads or fused" -m gpu -q --maxfail=12 > $OUT/r03c21_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/r03c21_pytest.txt )
tail -5 $OUT/r03c21_pytest.txt
A="--steps 20 --warmup 5 --no-cpu-baseline --no-also --no-alone --layers 8 --bf16-grads"
F="--steps 40 --warmup 5 --no-cpu-baseline --no-also --no-alone --forward-only --bf16"
run() { local tag=$1; shift; timeout 300 "$@" > $OUT/r03c21_$tag.json 2>/dev/null; python -c "import json;d=json.load(open('$OUT/r03c21_$tag.json'));print('$tag', d['value'], d['ms_per_step_median'])"; }
for rep in 1 2; do
  run new_$rep python bench.py $A
  CUNET_LIB_PATH=$ROOT/cu_net_amd/libcunet_hip_prev.so run prev_$rep python tools/bench_tuning.py $A
done
run new_fwd python bench.py $F
CUNET_LIB_PATH=$ROOT/cu_net_amd/libcunet_hip_prev.so run prev_fwd python tools/bench_tuning.py $F
