#!/bin/sh
# Timing experiments: build ablated variants of the library (results invalid) and time them.
# usage (on the GPU box): sh tools/ablate.sh "0 2 16 128"
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ablate
for A in ${1:-0 2 16 128}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DGTARS_ABLATE=$A $EXTRA -I include \
    -o gpurun_out/ablate/lib_$A.so -x c++ gtars_amd/csrc/host.cpp -x hip gtars_amd/csrc/api.hip \
    -x hip gtars_amd/csrc/kernels.hip -x hip gtars_amd/csrc/sort.hip -x hip gtars_amd/csrc/igd_sweep.hip -x hip gtars_amd/csrc/tokenize_lds.hip -lz 2>/dev/null
  # (the last -x must be hip: a trailing "-x c++" input makes the driver skip the HIP device link)
  echo "== ablate $A"
  GTARS_AMD_LIB=$PWD/gpurun_out/ablate/lib_$A.so CONFIGS=${CONFIGS:-512:0} SIZES=${SIZES:-64000000} python tools/kbench.py 2>&1 | grep -v amdgpu.ids
done
