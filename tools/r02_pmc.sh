#!/bin/sh
# SQ counters of the tokenizer at 64M queries (separate --pmc passes, no trace domains)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
D=$PWD/gpurun_out/pmc64
rm -rf $D; mkdir -p $D
export CONFIGS=1024:0:0 SIZES=${SIZES:-64000000}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $D/a -- python3 tools/kbench.py > $D/a.out 2> $D/a.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $D/b -- python3 tools/kbench.py > $D/b.out 2> $D/b.err
rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $D/c -- python3 tools/kbench.py > $D/c.out 2> $D/c.err
python3 - <<'PY'
import csv, glob, collections
for d in "abc":
    for f in glob.glob(f"gpurun_out/pmc64/{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_tok_lds" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print(d, k, sum(v) / len(v), len(v))
PY
tail -3 $D/a.err $D/b.err $D/c.err
find $D -name "*.db" -delete
