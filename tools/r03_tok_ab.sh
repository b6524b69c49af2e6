#!/bin/sh
# A/B of tokenizer builds / environment switches: tools/kbench.py on one re-used batch per size.
# VARIANTS: names of build/variants/lib_<name>.so, or env:<VAR>=<value> to run the head library with that variable set
cd "$(dirname "$0")/.."
export CONFIGS=1024:0:0 SIZES=${SIZES:-1000000,8000000,64000000,256000000}
for v in "" $VARIANTS; do
  unset GTARS_AMD_LIB
  case "$v" in
    env:*) kv=${v#env:}; export "$kv";;
    "") ;;
    *) export GTARS_AMD_LIB=$PWD/build/variants/lib_$v.so;;
  esac
  echo "== ${v:-head}"
  python3 tools/kbench.py 2>/dev/null | python3 -c "
import json,sys
print(' '.join('%d:%.1fus(%.3f)' % (d['nq'], d['us'], d['hbm_frac']) for d in map(json.loads, sys.stdin)))"
  case "$v" in env:*) unset "${kv%%=*}";; esac
done
