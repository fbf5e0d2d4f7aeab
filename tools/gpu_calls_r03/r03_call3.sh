#!/bin/bash
# Round-3 GPU call 3: fused head loss, fat pools, bf16 tap-split 3x3 forward -- correctness subset, then config 2 / 3 timings and
# a bf16 kernel trace.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_exact.py tests/test_gpu_parity.py tests/test_gpu_augment.py "tests/test_gpu_nodes.py::test_every_node_backward_full_width_bf16_gradient_tensors" \
    "tests/test_gpu_configs.py::test_config3_cu_net8_k68" tests/test_gpu_quant.py::test_quantised_input_train_step_matches_oracle -m gpu -q --maxfail=12 --durations=8 > $OUT/r03c3_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/r03c3_pytest.txt )
tail -12 $OUT/r03c3_pytest.txt
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-alone"
run() { local tag=$1; shift; timeout 300 "$@" > $OUT/r03c3_$tag.json 2>/dev/null; python -c "import json;d=json.load(open('$OUT/r03c3_$tag.json'));print('$tag', d['value'], d['ms_per_step_median'])"; }
run f32 $B
run bf16 $B --layers 8 --bf16-grads --steps 20
export CUNET_LIB_PATH=$ROOT/cu_net_amd/libcunet_hip_tuning.so
T="python tools/bench_tuning.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-alone --layers 8 --bf16-grads"
CUNET_B16_TS=0 run bf16_ts0 $T
CUNET_B16_TS=4 run bf16_ts4 $T
CUNET_B16_TS=12 CUNET_B16_TS_BPC=1 run bf16_ts12_bpc1 $T
CUNET_B16_TS=12 CUNET_B16_TS_BPC=3 run bf16_ts12_bpc3 $T
unset CUNET_LIB_PATH
cd /tmp
P3="python $ROOT/bench.py --layers 8 --bf16-grads --steps 5 --warmup 3 --no-cpu-baseline --no-also --no-alone"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/r03c3_tr_bf16 -o t -- $P3 > /dev/null 2> $OUT/r03c3_tr_bf16.err
cd $ROOT
f() { ls $OUT/$1/*kernel_trace.csv 2>/dev/null | head -1; }
python tools/trace_overlap.py "$(f r03c3_tr_bf16)" > $OUT/r03c3_overlap_bf16.txt 2>&1
python tools/trace_summary.py "$(f r03c3_tr_bf16)" 70 > $OUT/r03c3_bf16_by_grid.txt 2>&1
gzip -c "$(f r03c3_tr_bf16)" > $OUT/r03c3_trace_bf16.csv.gz
rm -rf $OUT/r03c3_tr_bf16
head -5 $OUT/r03c3_overlap_bf16.txt
