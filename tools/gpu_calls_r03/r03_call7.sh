#!/bin/bash
# Round-3 GPU call 7: second-generation AND-popcount forward -- exactness tests, config 5 with and without it.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_quant.py "tests/test_gpu_configs.py::test_config5_cu_net16_k16_bits_w1_quantised_inputs" -m gpu -q --maxfail=12 > $OUT/r03c7_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/r03c7_pytest.txt )
tail -6 $OUT/r03c7_pytest.txt
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-alone --layers 16 --class-num 16 --bits-w 1"
run() { local tag=$1; shift; timeout 300 "$@" > $OUT/r03c7_$tag.json 2> $OUT/r03c7_$tag.err; python -c "import json;d=json.load(open('$OUT/r03c7_$tag.json'));print('$tag', d['value'], d['ms_per_step_median'])"; }
run mfma $B
run popcount $B --popcount
grep -E "popcount|conv3x3_fwd|conv1x1_fwd " $OUT/r03c7_popcount.err $OUT/r03c7_mfma.err
CUNET_NO_SIDE_STREAM=1 run popcount_serial $B --popcount
grep -E "popcount|conv3x3_fwd" $OUT/r03c7_popcount_serial.err
