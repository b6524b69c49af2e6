#!/bin/sh
# full GPU suite + default bench line (what the driver runs at round end)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/suite_pytest.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/suite_bench.json 2> gpurun_out/suite_bench.err
cat gpurun_out/suite_pytest.txt; tail -c 6000 gpurun_out/suite_bench.json; tail -5 gpurun_out/suite_bench.err
