#!/bin/sh
# rocprofv3 evidence for round 6: kernel trace + stats of the bench command, FETCH_SIZE / WRITE_SIZE in separate
# --pmc passes (never combined with a trace domain), kernel trace + PMC passes of the IGD config-3 call chain.
# Run on the GPU box:  sh tools/profile_r06.sh ; python tools/collect_profiles.py gpurun_out/prof_r06 profiles/r06
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
D=$PWD/gpurun_out/prof_r06
rm -rf $D; mkdir -p $D
# (bench.py under rocprofv3 needs --no-pmc: it must not start child profilers from a profiled process)
BENCH="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-pmc --min-seconds 0.05"
rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- $BENCH > $D/bench_trace.json 2> $D/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch -- $BENCH > $D/bench_fetch.json 2> $D/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/write -- $BENCH > $D/bench_write.json 2> $D/write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $D/igd -- python3 tools/igd_bench.py > $D/igd.json 2> $D/igd.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/igd_fetch -- python3 bench.py --igd-pmc-child > /dev/null 2> $D/igd_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/igd_write -- python3 bench.py --igd-pmc-child > /dev/null 2> $D/igd_write.err
python3 - <<'PY'
import csv, glob, json, collections
D = "gpurun_out/prof_r06"
out = {}
for which, scale in (("igd_fetch", 2048.0), ("igd_write", 1024.0)):
    f = sorted(glob.glob(f"{D}/{which}/**/*counter_collection.csv", recursive=True))[-1]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("gtars::", "").strip()
        agg[n][0] += float(r["Counter_Value"]) * scale
        agg[n][1] += 1
    out[which] = {k: {"bytes": v[0], "dispatches": v[1]} for k, v in agg.items()}
CALLS = 6  # bench.py IGD_PMC_CALLS
path = ("k_igd_begin", "k_igd_call_init", "k_igd_prep", "k_igd_route", "k_ms_", "k_split_", "k_igd_chrom_segments", "k_igd_tile_ranges", "k_igd_sweep")
per = {}
for k in set(out["igd_fetch"]) | set(out["igd_write"]):
    if k.startswith(path):
        per[k] = round((out["igd_fetch"].get(k, {"bytes": 0})["bytes"] + out["igd_write"].get(k, {"bytes": 0})["bytes"]) / CALLS)
tot = sum(per.values())
alg = 12 * 10_000_000 + 16 * 50_000_000 + 8 * 1000
json.dump({"bytes_per_call": tot, "vs_algorithmic": tot / alg, "algorithmic_bytes": alg, "by_kernel_per_call": dict(sorted(per.items())),
           "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py --igd-pmc-child (two passes; 6 shuffled-batch calls, "
                      "pairwise and binary alternating; FETCH_SIZE x2 per the gfx950 note)"},
          open(f"{D}/traffic_igd_config3.json", "w"), indent=1)
print(open(f"{D}/traffic_igd_config3.json").read())
PY
sh tools/r06_igd_trace.sh prof_r06/igd_chain > /dev/null 2>&1
find $D -name "*kernel_trace.csv" -size +2M -delete
find $D -name "*.db" -delete
tail -c 300 $D/bench_trace.json
