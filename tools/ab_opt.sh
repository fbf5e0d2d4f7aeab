#!/bin/bash
# A planner option at several values on ONE box, interleaved (the shipped library):  ab_opt.sh <option> <reps> <v1> <v2> ...
OPT=$1; REPS=$2; shift 2
B="--steps 40 --warmup 5 --no-also --no-alone --no-cpu-baseline"
B8="--steps 20 --warmup 4 --no-also --no-alone --no-cpu-baseline --layers 8 --bf16 --bf16-grads"
val() { tail -n1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))" 2>/dev/null; }
for rep in $(seq $REPS); do for v in "$@"; do echo "f32  $OPT=$v: $(python bench.py $B --planner-opt $OPT=$v 2>/dev/null | val)"; done; done
for rep in $(seq $REPS); do for v in "$@"; do echo "bf16 $OPT=$v: $(python bench.py $B8 --planner-opt $OPT=$v 2>/dev/null | val)"; done; done
