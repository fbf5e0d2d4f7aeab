#!/bin/bash
# Round-6 planner-option sweep at HEAD, one box: img/s per setting (config 2 fp32, then config 3 bf16).
B="--steps 40 --warmup 5 --no-also --no-alone --no-cpu-baseline"
run() { local tag="$1"; shift; local v=$(timeout 120 python bench.py $B "$@" 2>/dev/null | tail -n1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))" 2>/dev/null); echo "$tag: $v"; }
run "default (1)"
run "wgrad_fork_group=1" --planner-opt wgrad_fork_group=1
run "wgrad_fork_group=3" --planner-opt wgrad_fork_group=3
run "wgrad_fork_group=4" --planner-opt wgrad_fork_group=4
run "default (2)"
run "wgrad3_max_splits=160" --planner-opt wgrad3_max_splits=160
run "wgrad3_max_splits=224" --planner-opt wgrad3_max_splits=224
run "wgrad3_max_splits=256" --planner-opt wgrad3_max_splits=256
run "heads_on_side=1" --planner-opt heads_on_side=1
run "heads_on_side=0" --planner-opt heads_on_side=0
run "default (3)"
B="--steps 20 --warmup 4 --no-also --no-alone --no-cpu-baseline --layers 8 --bf16 --bf16-grads"
run "bf16 L8 default (1)"
run "bf16 L8 wgrad_fork_group_bf16=4" --planner-opt wgrad_fork_group_bf16=4
run "bf16 L8 wgrad_fork_group_bf16=6" --planner-opt wgrad_fork_group_bf16=6
run "bf16 L8 wgrad_fork_group_bf16=12" --planner-opt wgrad_fork_group_bf16=12
run "bf16 L8 wgrad_fork_group_bf16=16" --planner-opt wgrad_fork_group_bf16=16
run "bf16 L8 wgrad3_max_splits_bf16=96" --planner-opt wgrad3_max_splits_bf16=96
run "bf16 L8 wgrad3_max_splits_bf16=160" --planner-opt wgrad3_max_splits_bf16=160
run "bf16 L8 wgrad3_max_splits_bf16=192" --planner-opt wgrad3_max_splits_bf16=192
run "bf16 L8 heads_on_side=0" --planner-opt heads_on_side=0
run "bf16 L8 default (2)"
