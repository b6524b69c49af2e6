#!/bin/sh
# same-box A/B of library variants on config 3 (tools/igd_bench.py): VARIANTS="name ..." (build/variants/lib_<name>.so)
cd "$(dirname "$0")/.."
for v in "" $VARIANTS; do
  if [ -n "$v" ]; then export GTARS_AMD_LIB=$PWD/build/variants/lib_$v.so; else unset GTARS_AMD_LIB; fi
  echo "== ${v:-current}"
  python3 tools/igd_bench.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('  config 3', {k:d[k]['ms'] for k in ('pairwise','binary','pairwise_sorted_input','binary_sorted_input')}, {k: round(v, 4) for k, v in d['kernels_ms'].items()})"
done
