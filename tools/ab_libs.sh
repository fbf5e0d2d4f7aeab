#!/bin/bash
# A/B of two builds of the library on ONE box (run from the repo root through gpurun):  ab_libs.sh <base.so> <new.so> [reps] [extra bench flags]
# Both go through tools/bench_tuning.py (bench.py itself only measures the shipped library); fp32 CU-Net-2 and bf16 CU-Net-8 lines.
BASE=$1; NEW=$2; REPS=${3:-3}; shift 3 || true
B="--steps 40 --warmup 5 --no-also --no-alone --no-cpu-baseline $*"
B8="--steps 20 --warmup 4 --no-also --no-alone --no-cpu-baseline --layers 8 --bf16 --bf16-grads $*"
val() { tail -n1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))" 2>/dev/null; }
for rep in $(seq $REPS); do
  echo "f32  base: $(CUNET_LIB_PATH=$PWD/$BASE python tools/bench_tuning.py $B 2>/dev/null | val)"
  echo "f32  new : $(CUNET_LIB_PATH=$PWD/$NEW python tools/bench_tuning.py $B 2>/dev/null | val)"
done
for rep in $(seq $REPS); do
  echo "bf16 base: $(CUNET_LIB_PATH=$PWD/$BASE python tools/bench_tuning.py $B8 2>/dev/null | val)"
  echo "bf16 new : $(CUNET_LIB_PATH=$PWD/$NEW python tools/bench_tuning.py $B8 2>/dev/null | val)"
done
