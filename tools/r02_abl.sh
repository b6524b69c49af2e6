#!/bin/sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/abl_svc.txt; : > $O
echo "== svc full" >> $O
GTARS_TOK_SVC=1 CONFIGS=1024:0:0 SIZES=8000000,64000000,256000000 timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids >> $O
for A in $ABL; do
echo "== svc ablate $A" >> $O
GTARS_TOK_SVC=1 GTARS_AMD_LIB=$PWD/build/variants/lib_abl$A.so CONFIGS=1024:0:0 SIZES=64000000 timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
