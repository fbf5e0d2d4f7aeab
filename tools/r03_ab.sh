#!/bin/bash
# A/B on ONE box: the round-2 tree (git archive 6e39baa under _r02_tree/, not committed) against the current tree, config 2 and config 3.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
A="--steps 60 --warmup 5 --no-cpu-baseline --no-also --no-alone"
for rep in 1 2; do
  (cd _r02_tree && python bench.py $A 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('r02 f32 ', d['value'], d['ms_per_step_median'])")
  python bench.py $A 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('r03 f32 ', d['value'], d['ms_per_step_median'])"
done
(cd _r02_tree && python bench.py $A --layers 8 --bf16-grads --steps 20 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('r02 bf16', d['value'], d['ms_per_step_median'])")
python bench.py $A --layers 8 --bf16-grads --steps 20 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('r03 bf16', d['value'], d['ms_per_step_median'])"
