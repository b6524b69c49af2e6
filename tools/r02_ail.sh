#!/bin/sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/ail.txt; : > $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host.py -x -q -m gpu 2>&1 | tail -8 >> $O
python - >> $O 2>&1 <<'PY'
import time, numpy as np, os, sys, torch
sys.path.insert(0, os.getcwd())
import gtars_amd
from gtars_amd import synth
dev = torch.device("cuda:0")
for ov in (0, 1):
    u = synth.make_universe(100_000, overlapping=bool(ov)); q = synth.make_queries(u, 1_000_000)
    for kind, name in ((0, "bits"), (1, "ailist")):
        ix = gtars_amd.OverlapIndex(u["chrom"], u["start"], u["end"], n_chrom=synth.N_CHROM, kind=kind)
        for rep in (1, 64):
            d = [torch.from_numpy(q[k].view(np.int32)).to(dev).repeat(rep) for k in ("chrom", "start", "end")]
            n = d[0].numel(); off = torch.empty(n + 1, dtype=torch.int64, device=dev); ids = torch.empty(n + 1024, dtype=torch.int32, device=dev)
            st = torch.cuda.current_stream().cuda_stream
            f = lambda s=False: ix.tokenize_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), n, off.data_ptr(), ids.data_ptr(), ids.numel(), st, sync=s)
            f(True); torch.cuda.synchronize()
            reps = 100 if rep == 1 else 5
            t = time.perf_counter()
            for _ in range(reps): f()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / reps
            print(f"universe {'C2prime' if ov else 'C2'} {name:6s} nq {n:9d}: {dt*1e6:9.1f} us  {n/dt/1e9:7.2f} Gq/s")
PY
cat $O
