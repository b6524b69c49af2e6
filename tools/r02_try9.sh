#!/bin/sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/try9.txt; : > $O
LABEL="count, as shipped" python tools/count_bench.py 2>&1 | grep -v amdgpu.ids >> $O
for v in 1 2 3; do
LABEL="count exp $v" GTARS_AMD_LIB=$PWD/build/variants/lib_cexp$v.so python tools/count_bench.py 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
