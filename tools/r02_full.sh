#!/bin/sh
# end-of-round evidence: whole GPU suite, smoke, the default bench line, profiles
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gputest_r02.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest_r02.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/smoke_r02.txt 2>&1
timeout 900 python bench.py > gpurun_out/bench_r02.json 2> gpurun_out/bench_r02.err
sh tools/profile_r02.sh > gpurun_out/profile_r02.log 2>&1
SIZES=64000000 sh tools/r02_pmc.sh > gpurun_out/pmc64.txt 2>&1
tail -3 gpurun_out/gputest_r02.txt; tail -2 gpurun_out/smoke_r02.txt; tail -c 3000 gpurun_out/bench_r02.json
