#!/bin/sh
# A/B on one box: bench.py's large batches with HEAD's library and a variant
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/ab.txt; : > $O
for rep in 1 2; do
for L in "" build/variants/lib_prepipe.so; do
echo "== lib '$L'" >> $O
GTARS_AMD_LIB=${L:+$PWD/$L} timeout 600 python bench.py --no-cpu-baseline --large 64000000,256000000 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('1M', d['roofline']['avg_kernel_ms'], [ (x['queries'], round(x['ms'],4)) for x in d['roofline_large']])" >> $O
done; done
cat $O
