#!/bin/sh
# SQ / LDS counters of the IGD config-3 call chain (separate --pmc passes, no trace domains); KERNELS = name filter
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
D=$PWD/gpurun_out/pmc_igd
rm -rf $D; mkdir -p $D
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $D/a -- python3 tools/igd_bench.py > $D/a.out 2> $D/a.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $D/b -- python3 tools/igd_bench.py > $D/b.out 2> $D/b.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d $D/c -- python3 tools/igd_bench.py > $D/c.out 2> $D/c.err
python3 - <<'PY'
import csv, glob, collections
for d in "abc":
    for f in glob.glob(f"gpurun_out/pmc_igd/{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "k_igd" in n or "k_ms_" in n or "k_split" in n:
                short = n.split("(")[0].replace("void ", "").replace("gtars::", "")[:40]
                agg[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print(d, k[0], k[1], round(sum(v) / len(v), 1), len(v))
PY
tail -2 $D/a.err $D/b.err $D/c.err
find $D -name "*.db" -delete
