#!/bin/bash
# Round-3 GPU call 1 (run from the repo root through gpurun): full GPU test suite, grid-barrier probe, the default bench line,
# a weight-gradient split sweep (does leaving CUs to the caller's stream help the overlap?), and kernel traces of the overlapped
# and serial step of configs 2 and 3 reduced by tools/trace_overlap.py.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --durations=15 > $OUT/r03c1_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/r03c1_pytest.txt )
tail -25 $OUT/r03c1_pytest.txt
timeout 120 tools/probes/grid_barrier_probe.bin > $OUT/r03c1_barrier.txt 2>&1
cat $OUT/r03c1_barrier.txt
timeout 600 python bench.py > $OUT/r03c1_bench.json 2> $OUT/r03c1_bench.err
tail -c 3000 $OUT/r03c1_bench.json
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-alone"
for s in 224 192 160 128; do
  timeout 200 $B --planner-opt wgrad3_max_splits=$s > $OUT/r03c1_split_$s.json 2>/dev/null
  python -c "import json;d=json.load(open('$OUT/r03c1_split_$s.json'));print('wgrad3_max_splits=$s', d['value'], d['ms_per_step_median'])"
done
cd /tmp
P="python $ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-also --no-alone"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/r03c1_tr_f32 -o t -- $P > /dev/null 2> $OUT/r03c1_tr_f32.err
CUNET_NO_SIDE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/r03c1_tr_f32_serial -o t -- $P > /dev/null 2> $OUT/r03c1_tr_f32_serial.err
P3="python $ROOT/bench.py --layers 8 --bf16-grads --steps 5 --warmup 3 --no-cpu-baseline --no-also --no-alone"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/r03c1_tr_bf16 -o t -- $P3 > /dev/null 2> $OUT/r03c1_tr_bf16.err
cd $ROOT
f() { ls $OUT/$1/*kernel_trace.csv 2>/dev/null | head -1; }
python tools/trace_overlap.py "$(f r03c1_tr_f32)" "$(f r03c1_tr_f32_serial)" > $OUT/r03c1_overlap_f32.txt 2>&1
python tools/trace_overlap.py "$(f r03c1_tr_bf16)" > $OUT/r03c1_overlap_bf16.txt 2>&1
head -12 $OUT/r03c1_overlap_f32.txt
# keep the merged-back payload small: the raw traces stay on the box except the two overlapped ones (compressed)
gzip -c "$(f r03c1_tr_f32)" > $OUT/r03c1_trace_f32.csv.gz
gzip -c "$(f r03c1_tr_bf16)" > $OUT/r03c1_trace_bf16.csv.gz
gzip -c "$(f r03c1_tr_f32_serial)" > $OUT/r03c1_trace_f32_serial.csv.gz
rm -rf $OUT/r03c1_tr_f32 $OUT/r03c1_tr_f32_serial $OUT/r03c1_tr_bf16 $OUT/r03c1_tr_bf16_serial
