#!/bin/sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/try4.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15 >> $O
for K in 0 1; do
  for Q in 4 2 1; do
    echo "== kernel $K qpt $Q" >> $O
    GTARS_TOK_KERNEL=$K CONFIGS=1024:0:$Q SIZES=1000000,16000000,64000000 timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids >> $O
  done
done
echo "== kernel 1 qpt 4, explicit ids" >> $O
GTARS_TOK_EXPLICIT_IDS=1 GTARS_NO_AFFINE_IDS=1 CONFIGS=1024:0:4 SIZES=1000000,64000000 timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids >> $O
echo "== kernel 1 qpt 4, no staging" >> $O
GTARS_TOK_STAGE=0 CONFIGS=1024:0:4 SIZES=1000000,64000000 timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids >> $O
echo "== stamped build qpt 4" >> $O
GTARS_AMD_LIB=$PWD/build/variants/lib_stamp.so CONFIGS=1024:0:4 SIZES=1000000,64000000 timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids >> $O
cat $O
