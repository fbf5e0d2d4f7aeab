#!/bin/bash
# Round-6 profiles (run from the repo root through gpurun):  profile_r06.sh <commit>
#   config 2 (CU-Net-2, fp32): rocprofv3 --kernel-trace --stats overlapped and with CUNET_NO_SIDE_STREAM=1, FETCH_SIZE / WRITE_SIZE passes
#   config 3 (CU-Net-8, bf16 storage): the same (kernel stats overlapped + serial, both PMC passes)
#   both: one SQ-counter pass with the side stream off (wave / wait / MFMA-busy / vector-memory cycles per kernel), reduced per kernel
#         class to the MFMA pipe utilisation bench.py reports as roofline.mfma_busy (tools/pmc_mfma_busy.py)
# Everything lands under gpurun_out/r06p_*; copy what should be judged into profiles/.
set -u
export CUNET_BENCH_NO_CLASS_EVENTS=1      # (kernel traces without bench.py's class events: their marker packets showed up as gaps around the dominant class in round 5's overlap tables)
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
COMMIT=${1:-unknown}
f() { ls $OUT/$1/*kernel_trace.csv 2>/dev/null | head -1; }
prof() {   # tag, bench args...
  local tag=$1; shift
  local B="python $ROOT/bench.py --no-cpu-baseline --no-also --no-alone $*"
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r06p_$tag -o t -- $B --steps 10 --warmup 3 > $OUT/r06p_${tag}_bench.json 2> $OUT/r06p_$tag.err
  CUNET_NO_SIDE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r06p_${tag}_serial -o t -- $B --steps 10 --warmup 3 > $OUT/r06p_${tag}_serial_bench.json 2> $OUT/r06p_${tag}_serial.err
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/r06p_${tag}_rd -o pmc -- $B --steps 3 --warmup 2 > /dev/null 2> $OUT/r06p_${tag}_rd.err
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/r06p_${tag}_wr -o pmc -- $B --steps 3 --warmup 2 > /dev/null 2> $OUT/r06p_${tag}_wr.err
  cd $ROOT
  cp "$(ls $OUT/r06p_$tag/*kernel_stats.csv | head -1)" $OUT/r06p_${tag}_kernel_stats.csv
  cp "$(ls $OUT/r06p_${tag}_serial/*kernel_stats.csv | head -1)" $OUT/r06p_${tag}_serial_kernel_stats.csv
  python tools/trace_summary.py "$(f r06p_$tag)" 70 > $OUT/r06p_${tag}_by_grid.txt 2>&1
  python tools/trace_summary.py "$(f r06p_${tag}_serial)" 70 > $OUT/r06p_${tag}_serial_by_grid.txt 2>&1
  python tools/trace_overlap.py "$(f r06p_$tag)" "$(f r06p_${tag}_serial)" > $OUT/r06p_${tag}_overlap.txt 2>&1
  python tools/step_timeline.py $OUT/r06p_$tag > $OUT/r06p_${tag}_timeline.txt 2>&1
  python tools/step_tail.py $OUT/r06p_$tag 34 > $OUT/r06p_${tag}_step_tail.txt 2>&1
}
prof f32
python tools/pmc_traffic.py $OUT/r06p_f32_rd $OUT/r06p_f32_wr $OUT/r06p_f32_traffic.json 2,68,24,f32 $COMMIT > $OUT/r06p_f32_traffic.txt 2>&1
prof bf16 --layers 8 --bf16-grads
python tools/pmc_traffic.py $OUT/r06p_bf16_rd $OUT/r06p_bf16_wr $OUT/r06p_bf16_traffic.json 8,68,24,bf16_grads $COMMIT > $OUT/r06p_bf16_traffic.txt 2>&1
# SQ counters per kernel (side stream off: every kernel alone), the longest 16 launches' classes of either workload
sq() {   # tag, bench args...
  local tag=$1; shift
  cd /tmp
  CUNET_NO_SIDE_STREAM=1 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM \
      --kernel-trace --output-format csv -d $OUT/r06p_${tag}_sq -o pmc -- python $ROOT/bench.py --no-cpu-baseline --no-also --no-alone "$@" --steps 2 --warmup 1 > /dev/null 2> $OUT/r06p_${tag}_sq.err
  cd $ROOT
  python tools/pmc_summary.py $OUT/r06p_${tag}_sq "" 16 > $OUT/r06p_${tag}_sq.txt 2>&1
  python tools/pmc_mfma_busy.py $OUT/r06p_${tag}_sq $OUT/r06p_${tag}_mfma_busy.json "$WKEY" $COMMIT > $OUT/r06p_${tag}_mfma_busy.txt 2>&1
  rm -rf $OUT/r06p_${tag}_sq
}
WKEY=2,68,24,f32 sq f32
WKEY=8,68,24,bf16_grads sq bf16 --layers 8 --bf16-grads
cat $OUT/r06p_f32_traffic.txt $OUT/r06p_bf16_traffic.txt
tail -1 $OUT/r06p_f32_bench.json | cut -c1-300; tail -1 $OUT/r06p_bf16_bench.json | cut -c1-300
rm -rf $OUT/r06p_f32 $OUT/r06p_f32_serial $OUT/r06p_f32_rd $OUT/r06p_f32_wr $OUT/r06p_bf16 $OUT/r06p_bf16_serial $OUT/r06p_bf16_rd $OUT/r06p_bf16_wr
