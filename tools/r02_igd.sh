#!/bin/sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/igd.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host.py -x -q -m gpu -k "igd or lola or config3 or config4" 2>&1 | tail -8 >> $O
python tools/igd_bench.py 2>&1 | grep -v amdgpu.ids >> $O
GTARS_IGD_FULL_SORT=1 python tools/igd_bench.py 2>&1 | grep -v amdgpu.ids >> $O
cat $O
