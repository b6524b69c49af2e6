#!/bin/sh
# Per-phase instruction counts: SQ counters of ablated builds (results invalid by construction).
# usage (on the GPU box): sh tools/pmc_ablate.sh "0 2 12 30"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_ablate; rm -rf $O; mkdir -p $O
export SIZES=${SIZES:-64000000} CONFIGS=${CONFIGS:-1024:0:4}
for A in ${1:-0 2 12 30}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DGTARS_ABLATE=$A $EXTRA -I include \
    -o $O/lib_$A.so -x c++ gtars_amd/csrc/host.cpp -x hip gtars_amd/csrc/api.hip \
    -x hip gtars_amd/csrc/kernels.hip -x hip gtars_amd/csrc/sort.hip -x hip gtars_amd/csrc/igd_sweep.hip -x hip gtars_amd/csrc/tokenize_lds.hip -lz 2>/dev/null
  export GTARS_AMD_LIB=$PWD/$O/lib_$A.so
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
     --output-format csv -d $O/a$A -- python3 tools/kbench.py > $O/a$A.log 2>&1
  rm -f $O/lib_$A.so
  python3 - $O/a$A $A <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "k_tok_lds" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("ablate",sys.argv[2],{k:round(sum(v)/len(v)/1e6,2) for k,v in sorted(acc.items())})
PY
  grep gqps $O/a$A.log
done
