#!/bin/sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/rounds.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 >> $O
for R in 1 2; do
echo "== rounds $R" >> $O
CONFIGS=1024:0:$R SIZES=1000000,8000000,64000000,256000000 timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
