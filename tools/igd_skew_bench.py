#!/usr/bin/env python3
"""Config 3 with a SKEWED query batch: a fraction of the 10M queries moved into one window of chr1 (what heavy-tile parts are
for: tiles that own far more queries than the average are served by several workgroups).  GTARS_IGD_NO_HEAVY_PARTS=1 for the
A/B.  Prints one line per shape."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import gtars_amd
from gtars_amd import synth

def main():
    F = 1000
    db = synth.make_igd_db(50_000_000, F, seed=6)
    g = gtars_amd.IgdIndex(db["chrom"], db["start"], db["end"], db["file"], n_chrom=synth.N_CHROM, n_files=F)
    q = synth.make_background_queries(10_000_000, seed=5)
    rng = np.random.default_rng(3)
    dev = torch.device("cuda:0")
    hits = torch.zeros(F, dtype=torch.int64, device=dev)
    for hot_frac, window in ((0.0, 0), (0.5, 50_000), (0.99, 50_000), (0.99, 5_000_000), (0.5, 1_000_000)):
        qq = {k: v.copy() for k, v in q.items()}
        n = len(qq["chrom"])
        if hot_frac:
            hot = rng.random(n) < hot_frac
            w = qq["end"] - qq["start"]
            qq["chrom"] = np.where(hot, 0, qq["chrom"]).astype(np.uint32)
            st = np.where(hot, rng.integers(50_000_000, 50_000_000 + window, n), qq["start"]).astype(np.uint32)
            qq["start"], qq["end"] = st, (st + w).astype(np.uint32)
        d = [torch.from_numpy(qq[k].view(np.int32)).to(dev) for k in ("chrom", "start", "end")]
        o = {"hot_fraction": hot_frac, "window_bp": window}
        for binary in (False, True):
            f = lambda: g.count_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), n, hits.data_ptr(), 1, binary, 0)
            f(); torch.cuda.synchronize()
            t = time.perf_counter(); f(); torch.cuda.synchronize()
            o["binary_ms" if binary else "pairwise_ms"] = round((time.perf_counter() - t) * 1e3, 3)
            o["binary_hits" if binary else "pairwise_hits"] = int(hits.sum())
        print(json.dumps(o), flush=True)

if __name__ == "__main__":
    main()
