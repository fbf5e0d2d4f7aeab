#!/bin/bash
# Builds libcunet_hip.so for gfx950 in-tree (cu_net_amd/libcunet_hip.so).
#   CUNET_TUNING=1 build.sh   -> cu_net_amd/libcunet_hip_tuning.so with -DCUNET_TUNING (environment knobs and the
#                                work-skipping timing switches compiled in; tools/ only, never the shipped library)
set -e
cd "$(dirname "$0")"
SRCS="augment_kernels.hip conv_kernels.hip wgrad_kernels.hip wgrad3_kernels.hip elementwise_kernels.hip quant_kernels.hip bf16_kernels.hip runtime.hip"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Wall -Wno-unused-function"
if [ -n "${CUNET_PROBE_DEFS:-}" ]; then      # probe build (tools only): the tuning library with extra -D switches, e.g. "-DCUNET_TEPI_WAVES=16"
  OUT=../libcunet_hip_probe.so; BUILD=build_probe; FLAGS="$FLAGS -DCUNET_TUNING $CUNET_PROBE_DEFS"
elif [ -n "${CUNET_PROBE_WG3_MIN_WAVES:-}" ]; then      # probe build (tools only): the fp32 wgrad3 kernels compiled for 3 waves per SIMD
  OUT=../libcunet_hip_probe.so; BUILD=build_probe; FLAGS="$FLAGS -DCUNET_TUNING -DCUNET_WG3_MIN_WAVES=$CUNET_PROBE_WG3_MIN_WAVES"
elif [ -n "${CUNET_TUNING:-}" ]; then
  OUT=../libcunet_hip_tuning.so; BUILD=build_tuning; FLAGS="$FLAGS -DCUNET_TUNING"
else
  OUT=../libcunet_hip.so; BUILD=build
fi
mkdir -p $BUILD
pids=()
objs=()
for f in $SRCS; do
  [ -f "$f" ] || continue
  hipcc $FLAGS -c $f -o $BUILD/${f%.hip}.o &
  pids+=($!)
  objs+=($BUILD/${f%.hip}.o)
done
hipcc $FLAGS -x hip -c plan.cpp -o $BUILD/plan.o &
pids+=($!)
objs+=($BUILD/plan.o)
for p in "${pids[@]}"; do wait $p; done
# link exactly the objects of this source list (a stale object of a removed file must not get in)
hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o $OUT
echo "built $(readlink -f $OUT)"
