#!/bin/bash
# Round-3 GPU call 2: the staged augment kernels and the exact quantised-loop tests, then sweeps of the two fork knobs.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_augment.py tests/test_gpu_quant.py tests/test_gpu_dp.py tests/test_gpu_exact.py "tests/test_gpu_nodes.py::test_whole_backward_is_the_sum_of_consumer_contributions" \
    tests/test_gpu_nodes.py::test_whole_backward_composition_full_width_cu_net4 -m gpu -q --maxfail=12 --durations=8 > $OUT/r03c2_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/r03c2_pytest.txt )
tail -15 $OUT/r03c2_pytest.txt
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-alone"
run() { # tag, args...
  local tag=$1; shift
  timeout 300 $B "$@" > $OUT/r03c2_$tag.json 2>/dev/null
  python -c "import json;d=json.load(open('$OUT/r03c2_$tag.json'));print('$tag', d['value'], d['ms_per_step_median'])"
}
for g in 1 2 4 8 16 64; do run f32_group$g --planner-opt wgrad_fork_group=$g; done
for w in 32 64 128; do run f32_fwdw$w --planner-opt fwd_fork_min_w=$w; done
run f32_group8_fwdw64 --planner-opt wgrad_fork_group=8 --planner-opt fwd_fork_min_w=64
for g in 1 4 8 16 64; do run bf16_group$g --layers 8 --bf16-grads --steps 20 --planner-opt wgrad_fork_group=$g; done
CUNET_NO_SIDE_STREAM=1 timeout 300 $B --layers 8 --bf16-grads --steps 20 > $OUT/r03c2_bf16_serial.json 2>/dev/null
python -c "import json;d=json.load(open('$OUT/r03c2_bf16_serial.json'));print('bf16 serial', d['value'], d['ms_per_step_median'])"
