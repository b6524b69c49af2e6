#!/bin/sh
# IGD parity subset + config 3
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/igd.txt; : > $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host.py -m gpu -x -q -k "igd or lola or config3 or config4 or dense or shard or rank" 2>&1 | tail -5 >> $O
for i in 1 2; do timeout 600 python tools/igd_bench.py 2>&1 | grep -v amdgpu.ids | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print({k:d[k]['ms'] for k in ('pairwise','binary','pairwise_sorted_input','binary_sorted_input')}, d['kernels_ms'])" >> $O; done
cat $O
