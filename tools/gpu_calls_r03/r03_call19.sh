#!/bin/bash
# Round-3 GPU call 19: host enqueue time of isolated steps; a whole step captured into a HIP graph against eager launches.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 python tools/host_enqueue_time.py | tee $OUT/r03c19_host_f32.json
timeout 200 python tools/host_enqueue_time.py --layers 8 --bf16-grads | tee $OUT/r03c19_host_bf16.json
timeout 300 python tools/graph_probe.py 2> $OUT/r03c19_graph_f32.err | tee $OUT/r03c19_graph_f32.json
timeout 300 python tools/graph_probe.py --layers 8 --bf16-grads --steps 20 2> $OUT/r03c19_graph_bf16.err | tee $OUT/r03c19_graph_bf16.json
tail -3 $OUT/r03c19_graph_f32.err
