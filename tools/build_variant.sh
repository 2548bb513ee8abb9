#!/bin/bash
# Timing-only variant of libfpn_hip.so: one chain unit (chain_bf16.hip, or FPN_VARIANT_UNIT=chain_reg)
# rebuilt with extra flags, the other objects as built.  Usage: bash tools/build_variant.sh <name> <flags...>  ->  tools/bin/libfpn_<name>.so
# (select it with FPN_LIB_PATH; FPN_VARIANT_BASE="" drops the accumulators-in-VGPRs option)
set -e
name=$1; shift
unit=${FPN_VARIANT_UNIT:-chain_bf16}
nonan=""
if [ "$unit" = chain_bf16 ]; then nonan="-fno-honor-nans -fno-slp-vectorize"; fi
cd "$(dirname "$0")/../3d-object-detection_amd/csrc"
mkdir -p ../../tools/bin build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -fhip-fp32-correctly-rounded-divide-sqrt \
    ${FPN_VARIANT_BASE--mllvm -amdgpu-mfma-vgpr-form} $nonan "$@" -c $unit.hip -o build/${unit}_$name.o
objs=$(ls build/*.o | grep -v "build/$unit")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libfpn_$name.so $objs build/${unit}_$name.o
rm -f build/${unit}_$name.o
