#!/bin/sh
# per-kernel durations of config 5's fused call (tools/r05_frag_trace.py under rocprofv3 --kernel-trace --stats), for the in-tree
# library and any variants given as arguments (build/variants/lib_*.so)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for v in "" "$@"; do
  if [ -z "$v" ]; then unset GTARS_AMD_LIB; else export GTARS_AMD_LIB=$PWD/$v; fi
  rm -rf gpurun_out/ft
  echo "== ${v:-in-tree}"
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ft -- python3 tools/r05_frag_trace.py 2>&1 | grep calls_ms
  python3 - <<'PY'
import csv, glob, re
f = glob.glob("gpurun_out/ft/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel + copy-kernel time in 6 calls: %.2f ms (%.2f per call)" % (tot / 1e6, tot / 6e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    m = re.search(r"(k_\w+(<[^>]*>)?|__amd_rocclr_\w+)", r["Name"])
    print("%-46s calls %5s avg_us %8.1f total_ms %7.2f" % ((m.group(1) if m else r["Name"][:46])[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
rm -rf gpurun_out/ft
