#!/bin/sh
# same-box A/B of config 3 (tools/igd_bench.py) and config 4 (tools/lola_bench.py): in-tree library vs a variant (GTARS_AMD_LIB)
# and environment switches; three alternating rounds each
cd "$(dirname "$0")/.."
V=${1:-build/variants/libhead.so}
pick() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
if 'pairwise' in d: print({k:d[k]['ms'] for k in ('pairwise','binary','pairwise_sorted_input','binary_sorted_input')})
else: print({'counts_ms':d['counts_ms']})
"; }
for r in 1 2 3; do
  echo "tree   c3 $(python3 tools/igd_bench.py 2>/dev/null | pick)"
  echo "var    c3 $(GTARS_AMD_LIB_OLDER=1 GTARS_AMD_LIB=$PWD/$V python3 tools/igd_bench.py 2>/dev/null | pick)"
done
for r in 1 2 3; do
  echo "tree shared      c4 $(python3 tools/lola_bench.py 2>/dev/null | pick)"
  echo "tree shared-u32  c4 $(GTARS_IGD_NO_B16=1 python3 tools/lola_bench.py 2>/dev/null | pick)"
  echo "tree set-by-set  c4 $(SET_BY_SET=1 python3 tools/lola_bench.py 2>/dev/null | pick)"
done
