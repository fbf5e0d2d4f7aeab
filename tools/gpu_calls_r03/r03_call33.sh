#!/bin/bash
# Round-3 GPU call 33: bf16 hand-over group default (8) -- bf16 tests + the config-3 line.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest "tests/test_gpu_configs.py::test_config3_cu_net8_k68" "tests/test_gpu_configs.py::test_config4_cu_net8_k16_rank_shard" tests/test_gpu_dp.py \
    "tests/test_gpu_exact.py::test_heads_on_the_side_stream_equal_heads_in_node_order" "tests/test_gpu_exact.py::test_adapter_pair_launches_equal_the_single_launches" -m gpu -q --maxfail=12 > $OUT/r03c33_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/r03c33_pytest.txt )
tail -3 $OUT/r03c33_pytest.txt
A="--steps 20 --warmup 5 --no-cpu-baseline --no-also --no-alone --layers 8 --bf16-grads"
run() { local tag=$1; shift; timeout 300 "$@" > $OUT/r03c33_$tag.json 2>/dev/null; python -c "import json;d=json.load(open('$OUT/r03c33_$tag.json'));print('$tag', d['value'], d['ms_per_step_median'])"; }
run bf16_default python bench.py $A
run bf16_g4 python bench.py $A --planner-opt wgrad_fork_group_bf16=4
run bf16_default2 python bench.py $A
