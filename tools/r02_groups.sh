#!/bin/sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/groups.txt; : > $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "geometries or helps_itself or full_size or config2 or insert_and_seek or holds_cus" 2>&1 | tail -5 >> $O
for L in "" build/variants/lib_nopf.so; do
for G in 1 2; do for R in 1 2; do
echo "== lib '$L' groups $G rounds $R" >> $O
GTARS_AMD_LIB=${L:+$PWD/$L} GTARS_TOK_GROUPS=$G CONFIGS=1024:0:$R SIZES=8000000,64000000,256000000 timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids >> $O
done; done; done
cat $O
