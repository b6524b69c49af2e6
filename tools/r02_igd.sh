#!/bin/sh
# IGD parity subset + config 3 (two-level LDS-reordered partition vs one-level)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/igd.txt; : > $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host.py -m gpu -x -q -k "igd or lola or config3 or config4 or dense or shard or rank" 2>&1 | tail -5 >> $O
timeout 600 python tools/igd_bench.py 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
GTARS_MS_ONE_LEVEL=1 timeout 600 python tools/igd_bench.py 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
cat $O
