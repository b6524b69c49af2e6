#!/bin/sh
# Build an experimental variant of the library (extra -D flags) into build/variants/lib_<name>.so.
# The variants travel to the GPU box with gpurun; select one with GTARS_AMD_LIB=$PWD/build/variants/lib_<name>.so
# usage: sh tools/build_variant.sh <name> "<extra hipcc flags>" [file.hip ...]
#   With file names: only those translation units are recompiled with the flags, the rest comes from build/obj (the objects
#   of the regular build: run `python __graft_entry__.py` first).  Without: every unit is recompiled.
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; shift 2 || true
mkdir -p build/variants build/vobj/$name
objs=""
for src in host.cpp lola_stats.cpp api.hip kernels.hip sort.hip igd_sweep.hip tokenize_lds.hip fragparse.hip inflate_dev.hip; do
  o=build/obj/$src.o
  if [ $# -eq 0 ] || echo " $* " | grep -q " $src "; then
    o=build/vobj/$name/$src.o
    case $src in *.hip) lang=hip;; *) lang=c++;; esac
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -I include -x $lang -c gtars_amd/csrc/$src -o $o &
  fi
  objs="$objs $o"
done
wait
hipcc --offload-arch=gfx950 -fPIC -shared -o build/variants/lib_$name.so $objs -lz -lpthread
echo "built build/variants/lib_$name.so"
