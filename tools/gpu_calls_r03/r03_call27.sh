#!/bin/bash
# Round-3 GPU call 27: anatomy of conv_bf16_kernel inside the CU-Net-8 bf16 TRAINING step (tuning build, CUNET_B16_DBG), per (kernel,
# grid) from rocprofv3 kernel traces.  32 no B preload, 64 no BN table, 256 no B reads, 512 no A loads after the first chunk,
# 1024 no MFMA, 2048 no output stores, 4096 no BN/ReLU arithmetic.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
export CUNET_LIB_PATH=$ROOT/cu_net_amd/libcunet_hip_tuning.so
cd /tmp
for d in 0 32 64 96 2048 5888 8032; do
  CUNET_B16_DBG=$d timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/r03c27_$d -o t -- python $ROOT/tools/bench_tuning.py --layers 8 --bf16-grads --steps 4 --warmup 2 --no-cpu-baseline --no-also --no-alone > /dev/null 2> $OUT/r03c27_$d.err
  python $ROOT/tools/trace_summary.py "$(ls $OUT/r03c27_$d/*kernel_trace.csv | head -1)" 200 | grep -E "conv_bf16_kernel<1|conv_bf16_pair_kernel<1" | head -12 > $OUT/r03c27_$d.txt
  rm -rf $OUT/r03c27_$d
  echo "== dbg $d"; head -8 $OUT/r03c27_$d.txt | cut -c1-150
done
