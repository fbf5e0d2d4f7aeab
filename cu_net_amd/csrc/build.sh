#!/bin/bash
# Builds libcunet_hip.so for gfx950 in-tree (cu_net_amd/libcunet_hip.so).
set -e
cd "$(dirname "$0")"
OUT=../libcunet_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Wall -Wno-unused-function"
mkdir -p build
pids=()
for f in conv_kernels.hip wgrad_kernels.hip elementwise_kernels.hip quant_kernels.hip bf16_kernels.hip runtime.hip; do
  hipcc $FLAGS -c $f -o build/${f%.hip}.o &
  pids+=($!)
done
hipcc $FLAGS -x hip -c plan.cpp -o build/plan.o &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o $OUT
echo "built $(readlink -f $OUT)"
