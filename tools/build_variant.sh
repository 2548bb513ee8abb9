#!/bin/bash
# Timing-only variant of libfpn_hip.so: chain_bf16.hip rebuilt with extra flags, the other
# objects as built.  Usage: bash tools/build_variant.sh <name> <flags...>  ->  tools/bin/libfpn_<name>.so
# (select it with FPN_LIB_PATH; FPN_VARIANT_BASE="" drops the accumulators-in-VGPRs option)
set -e
name=$1; shift
cd "$(dirname "$0")/../3d-object-detection_amd/csrc"
mkdir -p ../../tools/bin build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -fhip-fp32-correctly-rounded-divide-sqrt \
    ${FPN_VARIANT_BASE--mllvm -amdgpu-mfma-vgpr-form} -fno-honor-nans "$@" -c chain_bf16.hip -o build/chain_bf16_$name.o
objs=$(ls build/*.o | grep -v "chain_bf16")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libfpn_$name.so $objs build/chain_bf16_$name.o
rm -f build/chain_bf16_$name.o
