#!/bin/sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/try3.txt; : > $O
for Q in 4 1; do
echo "== stamped build qpt $Q" >> $O
GTARS_AMD_LIB=$PWD/build/variants/lib_stamp.so CONFIGS=1024:0:$Q SIZES=1000000,64000000 timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
