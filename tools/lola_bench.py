#!/usr/bin/env python3
"""BASELINE config 4 on ONE GPU: LOLA support counts, 1 user set + universe vs a 2,000-set region DB.
(The 8-GPU form shards the queries by range and all-reduces the F-long vectors: gtars_amd/sharding.py.)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import gtars_amd
from gtars_amd import synth
from gtars_amd._lib import check, lib

def main():
    F = int(os.environ.get("F", "2000")); per = int(os.environ.get("PER", "25000"))
    nuni = int(os.environ.get("NUNI", "1000000")); nuser = int(os.environ.get("NUSER", "100000"))
    dev = torch.device("cuda:0")
    db = synth.make_igd_db(F * per, F, seed=6)
    uni = synth.make_universe(nuni, seed=3)
    rng = np.random.default_rng(9)
    sel = np.sort(rng.choice(len(uni["chrom"]), nuser, replace=False))
    t = time.time(); g = gtars_amd.IgdIndex(db["chrom"], db["start"], db["end"], db["file"], n_chrom=synth.N_CHROM, n_files=F); tb = time.time() - t
    st = torch.cuda.current_stream().cuda_stream
    def dv(a): return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
    uq = [dv(uni[k]) for k in ("chrom", "start", "end")]
    sq = [dv(uni[k][sel]) for k in ("chrom", "start", "end")]
    uh = torch.zeros(F, dtype=torch.int64, device=dev); sh = torch.zeros(F, dtype=torch.int64, device=dev)
    cells = [torch.empty(F, dtype=torch.int64, device=dev) for _ in range(4)]
    def run():
        g.count_device(uq[0].data_ptr(), uq[1].data_ptr(), uq[2].data_ptr(), len(uni["chrom"]), uh.data_ptr(), 1, True, st)
        g.count_device(sq[0].data_ptr(), sq[1].data_ptr(), sq[2].data_ptr(), nuser, sh.data_ptr(), 1, True, st)
        check(lib.gtars_lola_contingency_device(sh.data_ptr(), uh.data_ptr(), F, nuser, len(uni["chrom"]), *[c.data_ptr() for c in cells], st))
    run(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    a, b, c, d = [x.cpu().numpy() for x in cells]
    ok = bool(((a + b) == uh.cpu().numpy()).all() and ((a + c) == nuser).all() and ((a + b + c + d) == len(uni["chrom"])).all())
    print(json.dumps({"F": F, "db_intervals": F * per, "universe": len(uni["chrom"]), "user": nuser, "build_s": round(tb, 2),
                      "counts_ms": round(dt * 1e3, 3), "identities_hold": ok, "support_sum": int(a.sum())}))
if __name__ == "__main__":
    main()
