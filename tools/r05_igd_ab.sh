#!/bin/sh
# same-box A/B of the sweep's forms on config 3 (tools/igd_bench.py): ENVS="VAR=1 ..." each run with that one switch set
cd "$(dirname "$0")/.."
for v in "" $ENVS; do
  echo "== ${v:-current}"
  env $v python3 tools/igd_bench.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('  config 3', {k:d[k]['ms'] for k in ('pairwise','binary','pairwise_sorted_input','binary_sorted_input')}, {k: round(v, 4) for k, v in d['kernels_ms'].items()})"
done
