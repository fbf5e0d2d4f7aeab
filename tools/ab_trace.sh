#!/bin/bash
# Per-(kernel, grid) durations of two builds under rocprofv3 --kernel-trace, serial (side stream off) so every kernel is alone:
#   ab_trace.sh <base.so> <new.so> <name filter regex> [bench flags]
BASE=$1; NEW=$2; FILT=$3; shift 3
ROOT=$(pwd); OUT=$ROOT/gpurun_out/abt; mkdir -p $OUT; export TMPDIR=/tmp
for tag in base new; do
  LIB=$BASE; [ $tag = new ] && LIB=$NEW
  cd /tmp
  CUNET_BENCH_NO_CLASS_EVENTS=1 CUNET_NO_SIDE_STREAM=1 CUNET_LIB_PATH=$ROOT/$LIB timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$tag -o t -- python $ROOT/tools/bench_tuning.py --no-cpu-baseline --no-also --no-alone --steps 6 --warmup 3 "$@" > /dev/null 2> $OUT/$tag.err
  cd $ROOT
  python tools/trace_summary.py "$(ls $OUT/tr_$tag/*/*kernel_trace.csv $OUT/tr_$tag/*kernel_trace.csv 2>/dev/null | head -1)" 400 > $OUT/sum_$tag.txt
  rm -rf $OUT/tr_$tag
done
python - "$FILT" <<'PY'
import re, sys
filt = re.compile(sys.argv[1])
def load(f):
    d = {}
    for line in open(f):
        p = line.split()
        if len(p) < 9 or not p[0].isdigit():
            continue
        # name may contain spaces: columns from the right are tot mn avg n wg gy gx
        tot, mn, avg, n, wg, gy, gx = p[-1], p[-2], p[-3], p[-4], p[-5], p[-6], p[-7]
        name = ' '.join(p[1:-7])
        d[(name, gx, wg)] = (int(n), float(avg), float(mn), float(tot))
    return d
a, b = load('gpurun_out/abt/sum_base.txt'), load('gpurun_out/abt/sum_new.txt')
ta = tb = 0.0
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0, 0, 0))[3])):
    if not filt.search(k[0]):
        continue
    x, y = a.get(k), b.get(k)
    print(f'{k[0][:52]:52s} grid {k[1]:>8s} wg {k[2]:>4s}  base n={x[0] if x else 0:4d} avg {x[1] if x else 0:7.2f} min {x[2] if x else 0:7.2f} | new n={y[0] if y else 0:4d} avg {y[1] if y else 0:7.2f} min {y[2] if y else 0:7.2f}')
    ta += x[3] if x else 0; tb += y[3] if y else 0
print(f'filtered total us: base {ta:.0f} new {tb:.0f}')
PY
