#!/bin/sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/groups.txt; : > $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 >> $O
for G in 1 2; do for R in 1 2; do
echo "== groups $G rounds $R" >> $O
GTARS_TOK_GROUPS=$G CONFIGS=1024:0:$R SIZES=1000000,8000000,64000000,256000000 timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids >> $O
done; done
cat $O
