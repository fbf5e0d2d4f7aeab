#!/bin/bash
# Where the stem planes kernel's time goes (tools only): probe builds with its loads / its contraction compiled out, every kernel alone on the GPU.
#   probe_stem_planes.sh <lib.so> ...
ROOT=$(pwd); OUT=$ROOT/gpurun_out/spl; mkdir -p $OUT; export TMPDIR=/tmp
for LIB in "$@"; do
  tag=$(basename $LIB .so)
  cd /tmp
  CUNET_BENCH_NO_CLASS_EVENTS=1 CUNET_NO_SIDE_STREAM=1 CUNET_LIB_PATH=$ROOT/$LIB timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$tag -o t -- python $ROOT/tools/bench_tuning.py --no-cpu-baseline --no-also --no-alone --steps 6 --warmup 3 --planner-opt stem_wgrad_planes=1 > /dev/null 2> $OUT/$tag.err
  cd $ROOT
  python tools/trace_summary.py "$(ls $OUT/tr_$tag/*/*kernel_trace.csv $OUT/tr_$tag/*kernel_trace.csv 2>/dev/null | head -1)" 400 > $OUT/sum_$tag.txt
  rm -rf $OUT/tr_$tag
  echo "== $tag"; grep -i "stem" $OUT/sum_$tag.txt | cut -c1-200
done
