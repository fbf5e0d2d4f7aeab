set -u
ROOT=$(pwd); export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also --no-alone"
cd /tmp
CUNET_NO_SIDE_STREAM=1 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmcsq_r02 -o pmc -- $BENCH > /dev/null 2> $ROOT/gpurun_out/pmcsq_r02.err
cd $ROOT
for pat in "conv_kernel<2" "wgrad3" "conv_kernel<0"; do python tools/pmc_summary.py gpurun_out/pmcsq_r02 "$pat" 8; done
