#!/bin/bash
B="--steps 60 --warmup 5 --no-also --no-alone --no-cpu-baseline"
run() { local tag="$1"; shift; local v=$(timeout 120 python bench.py $B "$@" 2>/dev/null | tail -n1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))" 2>/dev/null); echo "$tag: $v"; }
for rep in 1 2 3; do
run "fork_group=4 (default)"
run "fork_group=1"  --planner-opt wgrad_fork_group=1
run "fork_group=2"  --planner-opt wgrad_fork_group=2
run "fork_group=3"  --planner-opt wgrad_fork_group=3
done
run "fork_group=2 splits=160" --planner-opt wgrad_fork_group=2 --planner-opt wgrad3_max_splits=160
run "fork_group=2 splits=224" --planner-opt wgrad_fork_group=2 --planner-opt wgrad3_max_splits=224
run "fork_group=2 splits=256" --planner-opt wgrad_fork_group=2 --planner-opt wgrad3_max_splits=256
run "fork_group=2 heads_on_side=1" --planner-opt wgrad_fork_group=2 --planner-opt heads_on_side=1
run "fork_group=2 (again)"  --planner-opt wgrad_fork_group=2
B="--steps 10 --warmup 3 --no-also --no-alone --no-cpu-baseline --layers 16 --class-num 16 --bits-w 1"
run "config5 fork_group=4" 
run "config5 fork_group=2" --planner-opt wgrad_fork_group=2
run "config5 fork_group=3" --planner-opt wgrad_fork_group=3
run "config5 fork_group=4 (again)" 
