#!/bin/bash
# Variant library for A/B runs: tools/build_variant.sh NAME SOURCE.hip "-DFLAG=.. ..."  ->  fluidnet_amd/lib/variants/libfnet_NAME.so
# (one source recompiled with the extra flags, the other objects taken from the regular build; select with FNET_LIB=...)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; flags=$3
python -m fluidnet_amd.build > /dev/null
mkdir -p fluidnet_amd/lib/variants
obj=fluidnet_amd/lib/variants/${name}_${src%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-function -Wno-unused-value $flags -c fluidnet_amd/csrc/$src -o $obj
objs=""
for s in fnet_elementwise fnet_loss fnet_conv_ref fnet_conv_mfma fnet_optim fnet_plan fnet_rollout; do
  if [ "$s.hip" == "$src" ]; then objs="$objs $obj"; else objs="$objs fluidnet_amd/lib/$s.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o fluidnet_amd/lib/variants/libfnet_$name.so $objs
echo fluidnet_amd/lib/variants/libfnet_$name.so
