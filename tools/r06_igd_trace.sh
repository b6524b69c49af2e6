#!/bin/sh
# kernel trace of the IGD config-3 call chain WITH timestamps: per-kernel durations and the gaps between consecutive
# kernels of one call (what a graph / fewer launches could recover).  GTARS_AMD_LIB selects a variant library.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
D=$PWD/gpurun_out/${1:-igd_trace}
rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --output-format csv -d $D/t -- python3 tools/igd_bench.py > $D/igd.json 2> $D/igd.err
tail -c 1500 $D/igd.json
python3 - "$D" <<'PY'
import csv, glob, sys, collections
D = sys.argv[1]
f = sorted(glob.glob(f"{D}/t/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gtars::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
# calls = runs of kernels starting at k_igd_call_init
calls, cur = [], None
for s, e, n in rows:
    if n.startswith("k_igd_call_init") or n.startswith("k_igd_begin"):
        cur = []
        calls.append(cur)
    if cur is not None and (n.startswith("k_igd") or n.startswith("k_ms_") or n.startswith("k_split")):
        cur.append((s, e, n))
agg = collections.defaultdict(list)
for c in calls:
    if len(c) < 3: continue
    route = [e - s for s, e, n in c if n.startswith("k_split_pass")]
    sig = ("shuffled" if route and route[0] > 20000 else "in order",) + tuple(x[2][:28] for x in c)
    span = c[-1][1] - c[0][0]
    busy = sum(e - s for s, e, _ in c)
    agg[sig].append((span, busy, [e - s for s, e, _ in c], [c[i + 1][0] - c[i][1] for i in range(len(c) - 1)]))
with open(f"{D}/chain_summary.txt", "w") as out:
    for sig, v in agg.items():
        v = v[1:] if len(v) > 2 else v
        n = len(v)
        label, sig = sig[0], sig[1:]
        print(f"{label} chain x{n}: span {sum(x[0] for x in v)/n/1e3:.1f} us, kernels {sum(x[1] for x in v)/n/1e3:.1f} us", file=out)
        for i, name in enumerate(sig):
            d = sum(x[2][i] for x in v) / n / 1e3
            g = sum(x[3][i] for x in v) / n / 1e3 if i < len(sig) - 1 else 0.0
            print(f"   {name:30s} {d:8.1f} us   gap after {g:6.1f} us", file=out)
print(open(f"{D}/chain_summary.txt").read())
PY
find $D -name "*.db" -delete
find $D -name "*kernel_trace.csv" -size +4M -delete
