#!/usr/bin/env python3
"""BASELINE config 3: IGD batch query, synthetic intervals vs an indexed multi-file database."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import gtars_amd
from gtars_amd import synth

def main():
    ndb = int(os.environ.get("NDB", "50000000")); nq = int(os.environ.get("NQ", "10000000")); F = int(os.environ.get("F", "1000"))
    dev = torch.device("cuda:0")
    t = time.time(); db = synth.make_igd_db(ndb, F); q = synth.make_background_queries(nq); tgen = time.time() - t
    t = time.time(); g = gtars_amd.IgdIndex(db["chrom"], db["start"], db["end"], db["file"], n_chrom=synth.N_CHROM, n_files=F); tbuild = time.time() - t
    qc, qs, qe = (torch.from_numpy(q[k].view(np.int32)).to(dev) for k in ("chrom", "start", "end"))
    hits = torch.zeros(F, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    out = {"ndb": ndb, "nq": nq, "F": F, "gen_s": round(tgen, 2), "build_s": round(tbuild, 2)}
    for binary in (False, True):
        g.count_device(qc.data_ptr(), qs.data_ptr(), qe.data_ptr(), nq, hits.data_ptr(), 1, binary, st)
        torch.cuda.synchronize()
        reps = 3
        t = time.perf_counter()
        for _ in range(reps):
            g.count_device(qc.data_ptr(), qs.data_ptr(), qe.data_ptr(), nq, hits.data_ptr(), 1, binary, st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        byts = 12 * nq + 16 * ndb + 8 * F
        out["binary" if binary else "pairwise"] = {"ms": round(dt * 1e3, 3), "qps": round(nq / dt), "hbm_frac": round(byts / dt / 8e12, 5),
                                                   "total_hits": int(hits.sum())}
    # the same batch already in (chromosome, start) order, as a sorted BED file would deliver it: no device sort
    order = np.lexsort((q["start"], q["chrom"]))
    sc_, ss_, se_ = (torch.from_numpy(q[k][order].view(np.int32)).to(dev) for k in ("chrom", "start", "end"))
    for binary in (False, True):
        g.count_device(sc_.data_ptr(), ss_.data_ptr(), se_.data_ptr(), nq, hits.data_ptr(), 1, binary, st)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            g.count_device(sc_.data_ptr(), ss_.data_ptr(), se_.data_ptr(), nq, hits.data_ptr(), 1, binary, st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 3
        out[("binary" if binary else "pairwise") + "_sorted_input"] = {"ms": round(dt * 1e3, 3), "qps": round(nq / dt),
                                                                       "hbm_frac": round((12 * nq + 16 * ndb + 8 * F) / dt / 8e12, 5),
                                                                       "total_hits": int(hits.sum())}
    del sc_, ss_, se_
    from gtars_amd import _lib
    _lib.lib.gtars_prof_reset(); _lib.lib.gtars_prof_enable(1)
    for binary in (False, True):
        g.count_device(qc.data_ptr(), qs.data_ptr(), qe.data_ptr(), nq, hits.data_ptr(), 1, binary, st)
    torch.cuda.synchronize()
    out["kernels_ms"] = {k: round(v["total_ms"], 3) for k, v in _lib.prof_read().items()}
    _lib.lib.gtars_prof_enable(0)
    # parity at this size is asserted by tests/test_gpu_parity.py::test_config3_igd_full_size_properties
    print(json.dumps(out))
if __name__ == "__main__":
    main()
