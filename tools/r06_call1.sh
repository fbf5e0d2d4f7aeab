#!/bin/bash
# Round-6 call 1 (one box): hand-over event flags A/B on the tuning build + one-step time lines (train fp32, train bf16 L8, eval forward).
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/c1; mkdir -p $OUT; export TMPDIR=/tmp
CUNET_TUNING=1 bash cu_net_amd/csrc/build.sh > $OUT/build_tuning.log 2>&1 || { echo "tuning build failed"; tail -20 $OUT/build_tuning.log; }
B="--steps 40 --warmup 5 --no-also --no-alone --no-cpu-baseline"
val() { tail -n1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))" 2>/dev/null; }
run() { local tag="$1"; shift; echo "$tag: $( "$@" 2>/dev/null | val )"; }
T="python tools/bench_tuning.py"
for rep in 1 2; do
  run "f32 shipped            " python bench.py $B
  run "f32 tuning flags=0     " env CUNET_FORK_FLAGS=0 $T $B
  run "f32 tuning flags=1     " env CUNET_FORK_FLAGS=1 $T $B
  run "f32 tuning flags=2     " env CUNET_FORK_FLAGS=2 $T $B
  run "f32 tuning flags=6     " env CUNET_FORK_FLAGS=6 $T $B
  run "f32 tuning flags=7     " env CUNET_FORK_FLAGS=7 $T $B
done
B8="--steps 20 --warmup 4 --no-also --no-alone --no-cpu-baseline --layers 8 --bf16 --bf16-grads"
for rep in 1 2; do
  run "bf16 L8 tuning flags=0 " env CUNET_FORK_FLAGS=0 $T $B8
  run "bf16 L8 tuning flags=2 " env CUNET_FORK_FLAGS=2 $T $B8
  run "bf16 L8 tuning flags=6 " env CUNET_FORK_FLAGS=6 $T $B8
  run "bf16 L8 tuning flags=7 " env CUNET_FORK_FLAGS=7 $T $B8
done
# time lines
tl() { local tag=$1; shift
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$tag -o t -- python $ROOT/bench.py --no-cpu-baseline --no-also --no-alone --steps 6 --warmup 3 "$@" > $OUT/tr_${tag}_bench.json 2> $OUT/tr_$tag.err
  cd $ROOT
  python tools/step_timeline.py $OUT/tr_$tag > $OUT/timeline_$tag.txt 2>&1
  python tools/trace_overlap.py "$(ls $OUT/tr_$tag/*/*kernel_trace.csv $OUT/tr_$tag/*kernel_trace.csv 2>/dev/null | head -1)" "$(ls $OUT/tr_$tag/*/*kernel_trace.csv $OUT/tr_$tag/*kernel_trace.csv 2>/dev/null | head -1)" > $OUT/overlap_$tag.txt 2>&1
  rm -rf $OUT/tr_$tag
}
tl f32
tl bf16 --layers 8 --bf16 --bf16-grads
tl fwd --forward-only
ls -la $OUT | head -30
