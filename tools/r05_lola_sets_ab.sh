#!/bin/sh
# LOLA config 4 (tools/lola_bench.py): the universe and the user set swept as they arrive (each in order) against the partitioned form
# (GTARS_IGD_SETS_ALWAYS_PARTITION=1: what the library did before), same box
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for sw in "" 1; do
    if [ -z "$sw" ]; then unset GTARS_IGD_SETS_ALWAYS_PARTITION; echo "== sets in order: swept as they arrive"; else export GTARS_IGD_SETS_ALWAYS_PARTITION=1; echo "== always partitioned"; fi
    python3 tools/lola_bench.py 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in d if 'ms' in k or 'frac' in k})"
  done
done
