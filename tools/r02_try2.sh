#!/bin/sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/try2.txt; : > $O
for Q in 4 2 1; do
  echo "== wave kernel qpt $Q" >> $O
  CONFIGS=1024:0:$Q SIZES=1000000,16000000,64000000 timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids >> $O
done
echo "== stamped build qpt 4" >> $O
GTARS_AMD_LIB=$PWD/build/variants/lib_stamp.so CONFIGS=1024:0:4 SIZES=1000000,64000000 timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids >> $O
echo "== stamped build qpt 1" >> $O
GTARS_AMD_LIB=$PWD/build/variants/lib_stamp.so CONFIGS=1024:0:1 SIZES=1000000 timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids >> $O
cat $O
