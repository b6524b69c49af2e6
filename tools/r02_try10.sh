#!/bin/sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/try10.txt; : > $O
for PFV in 0 1; do
echo "== prefetch $PFV" >> $O
GTARS_TOK_PREFETCH=$PFV CONFIGS=1024:0:4 SIZES=1000000,16000000,64000000,256000000 timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids >> $O
done
GTARS_TOK_PREFETCH=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4 >> $O
cat $O
