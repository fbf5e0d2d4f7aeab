#!/bin/bash
# Round-6 call 2: one-step time lines WITHOUT bench.py's class events (CUNET_BENCH_NO_CLASS_EVENTS=1), fp32 L2 / bf16 L8 / serial fp32
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/c2; mkdir -p $OUT; export TMPDIR=/tmp
export CUNET_BENCH_NO_CLASS_EVENTS=1
tl() { local tag=$1; shift
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$tag -o t -- python $ROOT/bench.py --no-cpu-baseline --no-also --no-alone --steps 6 --warmup 3 "$@" > $OUT/tr_${tag}_bench.json 2> $OUT/tr_$tag.err
  cd $ROOT
  python tools/step_timeline.py $OUT/tr_$tag > $OUT/timeline_$tag.txt 2>&1
  F="$(ls $OUT/tr_$tag/*/*kernel_trace.csv $OUT/tr_$tag/*kernel_trace.csv 2>/dev/null | head -1)"
  python tools/trace_overlap.py "$F" "$F" > $OUT/overlap_$tag.txt 2>&1
  rm -rf $OUT/tr_$tag
}
tl f32
tl bf16 --layers 8 --bf16 --bf16-grads
CUNET_NO_SIDE_STREAM=1 tl f32_serial
ls -la $OUT | head -30
