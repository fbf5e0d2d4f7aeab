#!/bin/bash
# Round-3 GPU call 17: kernel traces of config 2 and config 3 at HEAD (timeline analysis: what sits on the caller's stream).
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
f() { ls $OUT/$1/*kernel_trace.csv 2>/dev/null | head -1; }
cd /tmp
for tag in f32 bf16; do
  if [ $tag = bf16 ]; then X="--layers 8 --bf16-grads"; else X=""; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/r03c17_tr_$tag -o t -- python $ROOT/bench.py $X --steps 6 --warmup 3 --no-cpu-baseline --no-also --no-alone > /dev/null 2> $OUT/r03c17_tr_$tag.err
  python $ROOT/tools/trace_overlap.py "$(f r03c17_tr_$tag)" > $OUT/r03c17_overlap_$tag.txt 2>&1
  gzip -c "$(f r03c17_tr_$tag)" > $OUT/r03c17_trace_$tag.csv.gz
  rm -rf $OUT/r03c17_tr_$tag
  head -6 $OUT/r03c17_overlap_$tag.txt
done
