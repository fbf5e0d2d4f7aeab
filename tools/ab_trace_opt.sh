#!/bin/bash
# Per-(kernel, grid) durations with a planner option at two values, under rocprofv3 --kernel-trace (overlapped step):
#   ab_trace_opt.sh <option> <v0> <v1> <name filter regex> [bench flags]
OPT=$1; V0=$2; V1=$3; FILT=$4; shift 4
ROOT=$(pwd); OUT=$ROOT/gpurun_out/abt; mkdir -p $OUT; export TMPDIR=/tmp
for v in $V0 $V1; do
  cd /tmp
  CUNET_BENCH_NO_CLASS_EVENTS=1 ${CUNET_ABT_ENV:-} timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$v -o t -- python $ROOT/bench.py --no-cpu-baseline --no-also --no-alone --steps 6 --warmup 3 --planner-opt $OPT=$v "$@" > /dev/null 2> $OUT/$v.err
  cd $ROOT
  python tools/trace_summary.py "$(ls $OUT/tr_$v/*/*kernel_trace.csv $OUT/tr_$v/*kernel_trace.csv 2>/dev/null | head -1)" 400 > $OUT/sum_$v.txt
  rm -rf $OUT/tr_$v
done
python - "$FILT" $V0 $V1 <<'PY'
import re, sys
filt = re.compile(sys.argv[1])
def load(f):
    d = {}
    for line in open(f):
        p = line.split()
        if len(p) < 9 or not p[0].isdigit():
            continue
        tot, mn, avg, n, wg, gy, gx = p[-1], p[-2], p[-3], p[-4], p[-5], p[-6], p[-7]
        d[(' '.join(p[1:-7]), gx, wg)] = (int(n), float(avg), float(mn), float(tot))
    return d
for v in sys.argv[2:4]:
    a = load(f'gpurun_out/abt/sum_{v}.txt')
    tot = 0.0
    print(f'== option value {v}')
    for k in sorted(a, key=lambda k: -a[k][3]):
        if filt.search(k[0]):
            n, avg, mn, t = a[k]
            print(f'  {k[0][:56]:56s} grid {k[1]:>8s} wg {k[2]:>4s} n={n:4d} avg {avg:7.2f} min {mn:7.2f} total {t:9.1f}')
            tot += t
    print(f'  filtered total us (9 steps): {tot:.0f}; all kernels: {sum(x[3] for x in a.values()):.0f}')
PY
