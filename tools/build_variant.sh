#!/bin/bash
# another build of the library for same-box A/B rounds: tools/build_variant.sh <name> [-DFLAG=..] ...  -> build/ab/lib_<name>.so
# (select it with PF_LIB=build/ab/lib_<name>.so; tools/abn.sh takes that as part of a setting)
set -e
name=$1; shift
cd "$(dirname "$0")/../pi-slam-fusion_amd/csrc"
mkdir -p ../../build/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-result "$@" -shared \
    -o ../../build/ab/lib_$name.so -x hip kernels.hip collapse_fused.hip single_band.hip fusion_map.cpp dist.cpp c_api.cpp image_io.cpp jpeg_decode.cpp png_decode.cpp jpeg_device.hip -lz -lpthread -ldl
echo built build/ab/lib_$name.so
