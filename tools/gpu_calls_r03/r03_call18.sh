#!/bin/bash
# Round-3 GPU call 18: heads on the side stream -- correctness subset, A/B against heads_on_side=0, host enqueue time.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_exact.py tests/test_gpu_dp.py "tests/test_gpu_configs.py::test_config3_cu_net8_k68" \
    tests/test_gpu_quant.py tests/test_gpu_parity.py -m gpu -q --maxfail=12 --durations=6 > $OUT/r03c18_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/r03c18_pytest.txt )
tail -12 $OUT/r03c18_pytest.txt
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-alone"
run() { local tag=$1; shift; timeout 300 "$@" > $OUT/r03c18_$tag.json 2>/dev/null; python -c "import json;d=json.load(open('$OUT/r03c18_$tag.json'));print('$tag', d['value'], d['ms_per_step_median'])"; }
for rep in 1 2; do
  run f32_side_$rep $B
  run f32_inline_$rep $B --planner-opt heads_on_side=0
  run bf16_side_$rep $B --layers 8 --bf16-grads --steps 20
  run bf16_inline_$rep $B --layers 8 --bf16-grads --steps 20 --planner-opt heads_on_side=0
done
run q_side $B --layers 16 --class-num 16 --bits-w 1 --steps 10
run q_inline $B --layers 16 --class-num 16 --bits-w 1 --steps 10 --planner-opt heads_on_side=0
python tools/host_enqueue_time.py | tee $OUT/r03c18_host_f32.json
python tools/host_enqueue_time.py --layers 8 --bf16-grads | tee $OUT/r03c18_host_bf16.json
