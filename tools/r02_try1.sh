#!/bin/sh
# first GPU contact of the wave-autonomous tokenizer: parity, then old vs new kernel timings
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/try1_pytest.txt
for K in 0 1; do
  for Q in 4 2 1; do
    [ "$K" = 0 ] && [ "$Q" = 1 ] && continue
    echo "== kernel $K qpt $Q" >> gpurun_out/try1_kbench.txt
    GTARS_TOK_KERNEL=$K CONFIGS=1024:0:$Q SIZES=1000000,16000000,64000000 timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/try1_kbench.txt
  done
done
cat gpurun_out/try1_pytest.txt gpurun_out/try1_kbench.txt
