#!/bin/sh
# Timing experiments: build ablated variants of the library (results invalid) and time them.
# usage (on the GPU box): sh tools/ablate.sh "0 1 2 4 8 16 32 12 28"
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ablate
for A in ${1:-0 1 2 4 8 16}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DGTARS_ABLATE=$A -I include \
    -o gpurun_out/ablate/lib_$A.so gtars_amd/csrc/api.hip gtars_amd/csrc/kernels.hip gtars_amd/csrc/tokenize_lds.hip -lz 2>/dev/null
  echo "== ablate $A"
  GTARS_AMD_LIB=$PWD/gpurun_out/ablate/lib_$A.so CONFIGS=${CONFIGS:-512:0} SIZES=${SIZES:-64000000} python tools/kbench.py 2>&1 | grep -v amdgpu.ids
done
