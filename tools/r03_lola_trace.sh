#!/bin/sh
# kernel trace of the config-4 count step, shared pass vs set by set (tools/lola_bench.py)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
D=$PWD/gpurun_out/lola_trace
rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D/shared -- python3 tools/lola_bench.py > $D/shared.out 2> $D/shared.err
SET_BY_SET=1 rocprofv3 --kernel-trace --stats --output-format csv -d $D/byset -- python3 tools/lola_bench.py > $D/byset.out 2> $D/byset.err
python3 - <<'PY'
import csv, glob
for d in ("shared", "byset"):
    for f in glob.glob(f"gpurun_out/lola_trace/{d}/**/*kernel_stats.csv", recursive=True):
        print("==", d)
        for r in csv.DictReader(open(f)):
            n = r["Name"]
            if any(k in n for k in ("k_igd", "k_ms_", "k_split", "k_lola")):
                print(f'{n.split("(")[0].replace("void ","").replace("gtars::","")[:44]:46s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"])/1e3:9.1f}')
PY
tail -1 $D/shared.out $D/byset.out
find $D -name "*.db" -delete
