#!/bin/bash
# another build of the library for same-box A/B rounds: tools/build_variant.sh <name> [-DFLAG=..] ...  -> tools/ab/lib_<name>.so
# (select it with PF_LIB=tools/ab/lib_<name>.so; tools/abn.sh takes that as part of a setting)
set -e
name=$1; shift
cd "$(dirname "$0")/../pi-slam-fusion_amd/csrc"
mkdir -p ../../tools/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-result "$@" -shared \
    -o ../../tools/ab/lib_$name.so -x hip kernels.hip single_band.hip fusion_map.cpp dist.cpp c_api.cpp image_io.cpp -lz -lpthread -ldl
echo built tools/ab/lib_$name.so
