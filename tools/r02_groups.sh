#!/bin/sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/groups.txt; : > $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "geometries or helps_itself or full_size or config2 or capacity or holds_cus or random_differential" 2>&1 | tail -5 >> $O
for i in 1 2; do
timeout 300 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 1M avg_kernel_ms', d['roofline']['avg_kernel_ms'], 'ms_per_step', d['ms_per_step'])" >> $O
done
CONFIGS=1024:0:0 SIZES=1000000,8000000,64000000,256000000 timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids >> $O
cat $O
