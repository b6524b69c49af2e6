#!/bin/sh
# Round 6, final library: the GPU suite, the soak fuzzers, the bench line as the driver runs it, the rocprofv3 evidence.
# Run on the GPU box: sh tools/r06_final.sh ; results under gpurun_out/r06/final/ (copied into profiles/r06/ by hand).
cd "$(dirname "$0")/.."
D=gpurun_out/r06/final
mkdir -p $D
git rev-parse HEAD > $D/head.txt 2>/dev/null
python -m pytest tests -m gpu -q 2>&1 | tail -8 > $D/gputest_full_suite.txt
(timeout 1500 python tests/soak/fuzz_tokenize.py 550 2>&1 | tail -1
 SORTED=1 timeout 600 python tests/soak/fuzz_tokenize.py 150 7000 2>&1 | tail -1
 timeout 1800 python tests/soak/fuzz_igd.py 1000 2>&1 | tail -1
 timeout 900 python tests/soak/fuzz_igd.py big 12 2>&1 | tail -1
 timeout 1500 python tests/soak/fuzz_fragments.py 900 2>&1 | tail -1) > $D/soak_summary.txt 2>&1
python bench.py > $D/bench_default_run.json 2> $D/bench_default_run.err
sh tools/profile_r06.sh > $D/profile_r06.log 2>&1
python tools/collect_profiles.py gpurun_out/prof_r06 $D > $D/collect.log 2>&1
cp gpurun_out/prof_r06/igd_chain/chain_summary.txt $D/igd_chain_summary.txt 2>/dev/null
cat $D/gputest_full_suite.txt $D/soak_summary.txt
tail -c 300 $D/bench_default_run.json
