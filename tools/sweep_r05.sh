#!/bin/bash
# Round-5 planner-option sweep at HEAD, one box (run from the repo root through gpurun): img/s of the CU-Net-2 fp32 train step per setting.
B="--steps 40 --warmup 5 --no-also --no-alone --no-cpu-baseline"
run() { local tag="$1"; shift; local v=$(timeout 120 python bench.py $B "$@" 2>/dev/null | tail -n1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))" 2>/dev/null); echo "$tag: $v"; }
run "default (1)"
run "dgrad_rows=768"            --planner-opt dgrad_rows=768
run "dgrad_rows=192"            --planner-opt dgrad_rows=192
run "wgrad3_max_splits=160"     --planner-opt wgrad3_max_splits=160
run "wgrad3_max_splits=224"     --planner-opt wgrad3_max_splits=224
run "wgrad_fork_group=2"        --planner-opt wgrad_fork_group=2
run "wgrad_fork_group=6"        --planner-opt wgrad_fork_group=6
run "wgrad_fork_group=8"        --planner-opt wgrad_fork_group=8
run "default (2)"
run "dgrad3_ring=192"           --planner-opt dgrad3_ring=192
run "conv3x3_ring_min_rows=128" --planner-opt conv3x3_ring_min_rows=128
run "heads_on_side=1"           --planner-opt heads_on_side=1
run "dgrad_nt=1"                --planner-opt dgrad_nt=1
run "pair_adapters=0"           --planner-opt pair_adapters=0
run "default (3)"
B="--steps 20 --warmup 4 --no-also --no-alone --no-cpu-baseline --layers 8 --bf16 --bf16-grads"
run "bf16 L8 default (1)"
run "bf16 L8 wgrad_fork_group_bf16=4"   --planner-opt wgrad_fork_group_bf16=4
run "bf16 L8 wgrad_fork_group_bf16=12"  --planner-opt wgrad_fork_group_bf16=12
run "bf16 L8 wgrad3_max_splits_bf16=96" --planner-opt wgrad3_max_splits_bf16=96
run "bf16 L8 wgrad3_max_splits_bf16=160" --planner-opt wgrad3_max_splits_bf16=160
run "bf16 L8 default (2)"
