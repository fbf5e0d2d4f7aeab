#!/bin/bash
# Round-3 GPU call 22: anatomy of conv_bf16_kernel (tuning build, CUNET_B16_DBG): which part of the kernel the time goes to.
# 256 no B reads from LDS, 512 no A loads after a tile's first chunk, 1024 no MFMA, 2048 no output stores, 4096 no BN/ReLU arithmetic
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
F="--steps 40 --warmup 5 --no-cpu-baseline --no-also --no-alone --forward-only --bf16"
for d in 0 256 512 1024 2048 4096 768 1792 3840 7936; do
  CUNET_B16_DBG=$d timeout 200 python tools/bench_tuning.py $F > $OUT/r03c22_dbg$d.json 2>/dev/null
  python -c "import json;d=json.load(open('$OUT/r03c22_dbg$d.json'));r=d['roofline'];print('dbg $d', d['value'], d['ms_per_step_median'], r['kernel'], r['launches'], r['avg_launch_us'])"
done
