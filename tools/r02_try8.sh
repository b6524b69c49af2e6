#!/bin/sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/try8.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -8 >> $O
echo "== qpt 4" >> $O
CONFIGS=1024:0:4 SIZES=1000000,16000000,64000000,256000000 timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids >> $O
echo "== qpt 4, no staging" >> $O
GTARS_TOK_STAGE=0 CONFIGS=1024:0:4 SIZES=1000000,64000000 timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids >> $O
echo "== qpt 4, explicit ids" >> $O
GTARS_NO_AFFINE_IDS=1 CONFIGS=1024:0:4 SIZES=1000000,64000000 timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids >> $O
echo "== qpt 4, sorted batch" >> $O
SORTED=1 CONFIGS=1024:0:4 SIZES=1000000,64000000 timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids >> $O
echo "== qpt 4, chip-like universe" >> $O
OVERLAP=1 CONFIGS=1024:0:4 SIZES=1000000,64000000 timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids >> $O
cat $O
