#!/usr/bin/env python3
"""AIList-order tokenization of a NESTED AIList index (an overlapping universe whose sub-list decomposition has several levels):
flat companion's LDS tokenizer + k_ailist_reorder (round 5) against the generic one-thread-per-query kernel
(GTARS_AILIST_NO_REORDER=1) and against Bits order on the same universe.  Prints kernel times from the library's profiling hooks."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, gtars_amd
from gtars_amd import _lib

rng = np.random.default_rng(1)
n, n_chrom, span = int(os.environ.get("NU", 100_000)), 3, int(os.environ.get("SPAN", 20_000_000))
c = rng.integers(0, n_chrom, n).astype(np.uint32)
s = rng.integers(0, span, n).astype(np.uint32)
w = np.where(rng.random(n) < 0.02, rng.integers(5_000, 100_000, n), rng.integers(100, 900, n))
e = (s + w).astype(np.uint32)
nq = int(os.environ.get("NQ", 16_000_000))
qc = rng.integers(0, n_chrom, nq).astype(np.uint32)
qs = rng.integers(0, span, nq).astype(np.uint32)
qe = (qs + rng.integers(1, 600, nq)).astype(np.uint32)
dev = torch.device("cuda:0")
d = [torch.from_numpy(x.view(np.int32)).to(dev) for x in (qc, qs, qe)]
st = torch.cuda.current_stream().cuda_stream
off = torch.empty(nq + 1, dtype=torch.int64, device=dev)
for kind, env, label in ((0, None, "Bits order"), (1, None, "AIList order: flat companion + reorder"), (1, "GTARS_AILIST_NO_REORDER", "AIList order: generic kernel")):
    if env:
        os.environ[env] = "1"
        gtars_amd.reload_env()
    ix = gtars_amd.OverlapIndex(c, s, e, n_chrom=n_chrom, kind=kind)
    h = ix.tokenize_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, off.data_ptr(), 0, 0, st)
    ids = torch.empty(h + 1024, dtype=torch.int32, device=dev)
    run = lambda: ix.tokenize_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, off.data_ptr(), ids.data_ptr(), ids.numel(), st)
    run()
    _lib.lib.gtars_prof_reset(); _lib.lib.gtars_prof_enable(1)
    for _ in range(5):
        run()
    p = {k: round(v["total_ms"] / max(v["launches"], 1), 4) for k, v in _lib.prof_read().items() if k.startswith("k_")}
    _lib.lib.gtars_prof_enable(0)
    sub = max(len(ix.sublist_offsets(ch)) for ch in range(n_chrom)) if kind else 0
    print(f"{label}: {nq} queries, {h} ids ({h / nq:.2f} per query), sub-lists per chromosome up to {sub}: kernel ms {p}, total {sum(p.values()):.4f}", flush=True)
    if env:
        del os.environ[env]
        gtars_amd.reload_env()
