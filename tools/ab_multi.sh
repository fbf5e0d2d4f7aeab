#!/bin/bash
# Several builds of the library on ONE box, interleaved:  ab_multi.sh <reps> <lib1.so> <lib2.so> ...   (through tools/bench_tuning.py)
REPS=$1; shift
B="--steps 40 --warmup 5 --no-also --no-alone --no-cpu-baseline"
B8="--steps 20 --warmup 4 --no-also --no-alone --no-cpu-baseline --layers 8 --bf16 --bf16-grads"
val() { tail -n1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))" 2>/dev/null; }
for rep in $(seq $REPS); do for L in "$@"; do echo "f32  $L: $(CUNET_LIB_PATH=$PWD/$L python tools/bench_tuning.py $B 2>/dev/null | val)"; done; done
for rep in $(seq $REPS); do for L in "$@"; do echo "bf16 $L: $(CUNET_LIB_PATH=$PWD/$L python tools/bench_tuning.py $B8 2>/dev/null | val)"; done; done
