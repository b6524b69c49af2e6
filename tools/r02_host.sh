#!/bin/sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/host.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_host.py -x -q -m gpu -k "tokenize_into" 2>&1 | tail -12 >> $O
python - >> $O 2>&1 <<'PY'
import time, statistics, numpy as np, os, sys
sys.path.insert(0, os.getcwd())
import gtars_amd
from gtars_amd import synth
u = synth.make_universe(100_000); q = synth.make_queries(u, 1_000_000)
ix = gtars_amd.OverlapIndex(u["chrom"], u["start"], u["end"], n_chrom=synth.N_CHROM)
out = (np.empty(1_000_001, dtype=np.uint64), np.empty(2_001_024, dtype=np.uint32))
for chunks in ("1", "2", "4", "8"):
    os.environ["GTARS_PIPE_CHUNKS"] = chunks
    ix.tokenize(q["chrom"], q["start"], q["end"], out=out)
    ts = []
    for _ in range(15):
        t = time.perf_counter(); ix.tokenize(q["chrom"], q["start"], q["end"], out=out); ts.append(time.perf_counter() - t)
    print("chunks", chunks, "median us", round(statistics.median(ts) * 1e6, 1), "q/s %.3g" % (1e6 / statistics.median(ts)))
del os.environ["GTARS_PIPE_CHUNKS"]
ts = []
for _ in range(9):
    t = time.perf_counter(); ix.tokenize(q["chrom"], q["start"], q["end"]); ts.append(time.perf_counter() - t)
print("allocating: median us", round(statistics.median(ts) * 1e6, 1), "q/s %.3g" % (1e6 / statistics.median(ts)))
PY
cat $O
