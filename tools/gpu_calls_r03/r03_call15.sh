#!/bin/bash
# Round-3 GPU call 15: adapter pairs (ahead + skip in one launch) -- correctness subset, then A/B against pair_adapters=0.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_exact.py tests/test_gpu_nodes.py "tests/test_gpu_configs.py::test_config3_cu_net8_k68" \
    tests/test_gpu_quant.py tests/test_gpu_parity.py -m gpu -q --maxfail=12 --durations=6 > $OUT/r03c15_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/r03c15_pytest.txt )
tail -15 $OUT/r03c15_pytest.txt
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-alone"
run() { local tag=$1; shift; timeout 300 "$@" > $OUT/r03c15_$tag.json 2>/dev/null; python -c "import json;d=json.load(open('$OUT/r03c15_$tag.json'));print('$tag', d['value'], d['ms_per_step_median'])"; }
for rep in 1 2; do
  run f32_pair_$rep $B
  run f32_single_$rep $B --planner-opt pair_adapters=0
  run bf16_pair_$rep $B --layers 8 --bf16-grads --steps 20
  run bf16_single_$rep $B --layers 8 --bf16-grads --steps 20 --planner-opt pair_adapters=0
done
run q_pair $B --layers 16 --class-num 16 --bits-w 1 --steps 10
run q_single $B --layers 16 --class-num 16 --bits-w 1 --steps 10 --planner-opt pair_adapters=0
