#!/bin/sh
# rocprofv3 evidence for the round: kernel trace + stats of the bench command, FETCH_SIZE / WRITE_SIZE in separate
# --pmc passes (never combined with a trace domain), kernel trace of the IGD bench.  Run on the GPU box:
#   sh tools/profile_r02.sh ; python tools/collect_profiles.py gpurun_out/prof_r02 profiles/r02
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
D=$PWD/gpurun_out/prof_r02
rm -rf $D; mkdir -p $D
BENCH="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --min-seconds 0.05"
rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- $BENCH > $D/bench_trace.json 2> $D/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch -- $BENCH > $D/bench_fetch.json 2> $D/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/write -- $BENCH > $D/bench_write.json 2> $D/write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $D/igd -- python3 tools/igd_bench.py > $D/igd.json 2> $D/igd.err
# keep only the small summaries (the raw traces are large)
find $D -name "*kernel_trace.csv" -size +2M -delete
find $D -name "*.db" -delete
ls -R $D | head -50
tail -c 400 $D/bench_trace.json
