#!/bin/sh
# Build an experimental variant of the library (extra -D flags) into build/variants/lib_<name>.so.
# The variants travel to the GPU box with gpurun; select one with GTARS_AMD_LIB=$PWD/build/variants/lib_<name>.so
# usage: sh tools/build_variant.sh <name> "<extra hipcc flags>"
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $2 -I include -o build/variants/lib_$1.so \
  -x c++ gtars_amd/csrc/host.cpp -x hip gtars_amd/csrc/api.hip -x hip gtars_amd/csrc/kernels.hip \
  -x hip gtars_amd/csrc/sort.hip -x hip gtars_amd/csrc/igd_sweep.hip -x hip gtars_amd/csrc/tokenize_lds.hip -lz -lpthread
echo "built build/variants/lib_$1.so"
