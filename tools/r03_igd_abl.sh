#!/bin/sh
# sweep ablations (tools/build_variant.sh igdabl<N> "-DGTARS_IGD_ABLATE=N"): kernel times of tools/igd_bench.py per variant
cd "$(dirname "$0")/.."
for v in "" $VARIANTS; do
  if [ -n "$v" ]; then export GTARS_AMD_LIB=$PWD/build/variants/lib_$v.so; else unset GTARS_AMD_LIB; fi
  echo "== ${v:-full}"
  python3 tools/igd_bench.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print({k:d[k]['ms'] for k in ('pairwise','binary','pairwise_sorted_input','binary_sorted_input')}, d['kernels_ms'])"
done
