#!/usr/bin/env python3
"""Config 3 with long records in the database (what the pieces view is for): FRAC of the 5e7 records get a width of
U[5000, WMAX) instead of 200 + U[0, 800).  Times the min_overlap == 1 batch counts with and without the pieces view
(GTARS_IGD_NO_PIECES=1 at build).  usage: python tools/igd_wide_bench.py [FRAC] [WMAX]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import gtars_amd
from gtars_amd import synth

def main():
    frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
    wmax = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
    F = 1000
    db = synth.make_igd_db(50_000_000, F, seed=6)
    rng = np.random.default_rng(1)
    n = len(db["start"])
    wide = rng.random(n) < frac
    db["end"] = np.where(wide, db["start"].astype(np.int64) + rng.integers(5_000, wmax, n), db["end"]).astype(db["end"].dtype)
    q = synth.make_background_queries(10_000_000, seed=5)
    dev = torch.device("cuda:0")
    d = [torch.from_numpy(q[k].view(np.int32)).to(dev) for k in ("chrom", "start", "end")]
    out = {"db_records": n, "long_fraction": frac, "long_width_max": wmax, "queries": len(q["chrom"])}
    ref = {}
    for label, env in (("pieces_view", None), ("flat_layout", "1")):
        if env: os.environ["GTARS_IGD_NO_PIECES"] = env
        else: os.environ.pop("GTARS_IGD_NO_PIECES", None)
        gtars_amd.reload_env()
        t = time.time()
        g = gtars_amd.IgdIndex(db["chrom"], db["start"], db["end"], db["file"], n_chrom=synth.N_CHROM, n_files=F)
        tb = time.time() - t
        hits = torch.zeros(F, dtype=torch.int64, device=dev)
        o = {"build_s": round(tb, 2)}
        for binary in (False, True):
            g.count_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), len(q["chrom"]), hits.data_ptr(), 1, binary, 0)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(3):
                g.count_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), len(q["chrom"]), hits.data_ptr(), 1, binary, 0)
            torch.cuda.synchronize()
            key = "binary" if binary else "pairwise"
            o[key + "_ms"] = round((time.perf_counter() - t) / 3 * 1e3, 3)
            h = hits.cpu().numpy().copy()
            if key in ref:
                o[key + "_same_as_pieces_view"] = bool(np.array_equal(h, ref[key]))
            else:
                ref[key] = h
            o[key + "_hits"] = int(h.sum())
        out[label] = o
        del g
        torch.cuda.empty_cache()
    print(json.dumps(out), flush=True)

if __name__ == "__main__":
    main()
