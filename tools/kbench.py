#!/usr/bin/env python3
"""Kernel iteration harness: times the fused tokenizer at several batch sizes / launch parameters."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import gtars_amd
from gtars_amd import synth, _lib

def run(ix, q, n, reps, dev):
    rep = max(n // len(q["chrom"]), 1)
    big = {k: torch.from_numpy(np.tile(q[k], rep).view(np.int32)).to(dev) for k in ("chrom", "start", "end")}
    n = len(q["chrom"]) * rep
    off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    ids = torch.empty(int(os.environ.get("IDS_PER_QUERY", "1")) * n + 1024, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    f = lambda s=False: ix.tokenize_device(big["chrom"].data_ptr(), big["start"].data_ptr(), big["end"].data_ptr(), n,
                                           off.data_ptr(), ids.data_ptr(), ids.numel(), st, sync=s)
    h = f(True)
    for _ in range(3): f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    return n, h, dt

def main():
    dev = torch.device("cuda:0")
    nu = int(os.environ.get("NU", "100000"))
    u = synth.make_universe(nu, overlapping=bool(int(os.environ.get("OVERLAP", "0"))))
    q = synth.make_queries(u, 1_000_000)
    widen = int(os.environ.get("WIDEN", "0"))  # hit-heavy batches: every query widened to this many bp (1 Mbp: ~33 ids per query)
    if widen:
        q["end"] = (q["start"].astype(np.int64) + widen).clip(max=0x7FFFFFFF).astype(q["end"].dtype)
    if int(os.environ.get("SORTED", "0")):  # position-sorted batch (a sorted BED / fragment file): neighbours share records
        order = np.lexsort((q["start"], q["chrom"]))
        q = {k: v[order] for k, v in q.items()}
    kind = int(os.environ.get("KIND", "0"))  # 1: AIList order (with OVERLAP=1: nested sub-lists -> flat companion + reorder)
    ix = gtars_amd.OverlapIndex(u["chrom"], u["start"], u["end"], n_chrom=synth.N_CHROM, kind=kind)
    sizes = [int(x) for x in os.environ.get("SIZES", "1000000,16000000,64000000").split(",")]
    configs = [c for c in os.environ.get("CONFIGS", "512:0:4").split(",")]
    for cfg in configs:
        tpb, wg, rr = (cfg.split(":") + ["0", "0"])[:3]
        os.environ["GTARS_TOK_QPT"] = rr
        os.environ["GTARS_TOK_ROUNDS"] = rr
        os.environ["GTARS_TOK_TPB"] = tpb
        os.environ["GTARS_TOK_WG_PER_CU"] = wg
        gtars_amd.reload_env()
        for n in sizes:
            reps = max(3, min(200, int(4e8 // n)))
            n2, h, dt = run(ix, q, n, reps, dev)
            byts = 12 * n2 + 8 * (n2 + 1) + 4 * h + 12 * len(u["chrom"])
            print(json.dumps({"tpb": tpb, "wg_per_cu": wg, "rounds": rr, "nq": n2, "us": round(dt * 1e6, 2), "gqps": round(n2 / dt / 1e9, 2), "ids": int(h), "g_ids_per_s": round(h / dt / 1e9, 1),
                              "hbm_frac": round(byts / dt / 8e12, 4)}), flush=True)
if __name__ == "__main__":
    main()
