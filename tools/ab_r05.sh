#!/bin/bash
# Round-5 A/B on one box (run from the repo root through gpurun): shipped library vs the probe build (-DCUNET_GATHER_FLAT: round-4 gather),
# row-tile data gradient v2 vs v1 (planner option dgrad_rows_v).  Prints img/s and the per-class times of one profiled step.
OUT=gpurun_out/ab_r05; mkdir -p $OUT
B="--steps 40 --warmup 5 --no-also --no-alone --no-cpu-baseline"
run() {   # tag, env-prefix..., -- args
  local tag=$1; shift
  ( "$@" ) > $OUT/$tag.json 2> $OUT/$tag.err
  python - "$tag" "$OUT/$tag.json" "$OUT/$tag.err" <<'PY'
import json, sys, re
tag, j, e = sys.argv[1:4]
try:
    d = json.loads(open(j).read().strip().splitlines()[-1])
    v = f"{d['value']:8.1f} img/s {d['ms_per_step']:7.3f} ms"
except Exception as ex:
    v = 'FAILED ' + repr(ex)
cls = {}
for line in open(e):
    m = re.match(r'\s+(\S+)\s+launches=\s*(\d+) ms=\s*([\d.]+)', line)
    if m and m.group(1) not in cls:
        cls[m.group(1)] = float(m.group(3))
keys = ['bn_bwd_apply', 'conv1x1_bwd_data', 'conv1x1_bwd_data_bf16', 'conv3x3_bwd_data', 'conv1x1_fwd', 'conv1x1_bwd_weight']
print(f'{tag:28s} {v}   ' + ' '.join(f'{k}={cls[k]:.3f}' for k in keys if k in cls))
PY
}
PROBE="env CUNET_LIB_PATH=$PWD/cu_net_amd/libcunet_hip_probe.so python tools/bench_tuning.py"
for rep in 1 2; do
  run f32_new_v2_$rep      python bench.py $B
  run f32_new_v1_$rep      python bench.py $B --planner-opt dgrad_rows_v=1
  run f32_flat_v1_$rep     $PROBE $B --planner-opt dgrad_rows_v=1
  run f32_flat_v2_$rep     $PROBE $B --planner-opt dgrad_rows_v=2
done
run bf16_new_1   python bench.py $B --layers 8 --bf16 --bf16-grads
run bf16_flat_1  $PROBE $B --layers 8 --bf16 --bf16-grads
run bf16_new_2   python bench.py $B --layers 8 --bf16 --bf16-grads
run bf16_flat_2  $PROBE $B --layers 8 --bf16 --bf16-grads
