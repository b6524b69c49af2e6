#!/bin/sh
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
D=$PWD/gpurun_out/valu
rm -rf $D; mkdir -p $D
export CONFIGS=1024:0:0 SIZES=64000000
for A in 0 8 24 6 7; do
if [ $A = 0 ]; then unset GTARS_AMD_LIB; else export GTARS_AMD_LIB=$PWD/build/variants/lib_abl$A.so; fi
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $D/a$A -- python3 tools/kbench.py > $D/a$A.out 2> $D/a$A.err
done
python3 - <<'PY'
import csv, glob, collections
for A in (0, 8, 24, 6, 7):
    for f in glob.glob(f"gpurun_out/valu/a{A}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_tok_lds" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(A, {k: round(sum(v) / len(v) / 1e6, 2) for k, v in sorted(agg.items())})
PY
find $D -name "*.db" -delete
