#!/usr/bin/env python3
"""Round 6: the sweep form of the tokenizer (k_tok_sweep, GTARS_TOK_SORTED) against the default kernel on batches in
(chromosome, start) order -- 1M queries (8 rotating batches) and 64M (the 1M batch tiled 64x and sorted on the device), universes
of 100k / 200k / 1M regions.  Both kernels' offsets and ids are compared on every batch."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import gtars_amd
from gtars_amd import synth


def timed(f, reps):
    st = torch.cuda.current_stream()
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        f()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def main():
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    sizes = [int(x) for x in os.environ.get("NUS", "100000,200000,1000000").split(",")]
    big_rep = int(os.environ.get("BIG_REP", "64"))
    for nu in sizes:
        u = synth.make_universe(nu)
        ix = gtars_amd.OverlapIndex(u["chrom"], u["start"], u["end"], n_chrom=synth.N_CHROM)
        nb = 8
        batches = []
        for b in range(nb):
            q = synth.make_queries(u, 1_000_000, seed=4 + 7919 * b)
            o = np.lexsort((q["start"], q["chrom"]))
            batches.append([torch.from_numpy(np.ascontiguousarray(q[k][o]).view(np.int32)).to(dev) for k in ("chrom", "start", "end")])
        nq = 1_000_000
        off = [torch.empty(nq + 1, dtype=torch.int64, device=dev) for _ in range(2)]
        ids = [torch.empty(2 * nq, dtype=torch.int32, device=dev) for _ in range(2)]
        res = {"universe": nu}
        for name, hint, k in (("default", ix.TOK_NARROW, 0), ("sweep", ix.TOK_NARROW | ix.TOK_SORTED, 1)):
            state = {"i": 0}

            def step():
                d = batches[state["i"] % nb]
                state["i"] += 1
                ix.tokenize_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, off[k].data_ptr(), ids[k].data_ptr(), ids[k].numel(),
                                   st, sync=False, hint=hint)

            dt = timed(step, 400)
            d = batches[0]
            h = ix.tokenize_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, off[k].data_ptr(), ids[k].data_ptr(), ids[k].numel(),
                                   st, sync=True, hint=hint)
            byts = 12 * nq + 8 * (nq + 1) + 4 * h + 12 * nu
            res[name + "_1M"] = {"us": round(dt * 1e6, 2), "frac": round(byts / dt / 8e12, 4), "hits": h}
        res["same_1M"] = bool(torch.equal(off[0], off[1]) and torch.equal(ids[0][:h], ids[1][:h]))
        del off, ids
        # 64M in order: the 1M batch tiled and sorted on the device by (chromosome, start)
        q = batches[0]
        big = [t.repeat(big_rep) for t in q]
        key = (big[0].to(torch.int64) & 0xFFFFFFFF) << 32 | (big[1].to(torch.int64) & 0xFFFFFFFF)
        order = torch.argsort(key, stable=True)
        del key
        big = [t[order].contiguous() for t in big]
        del order
        n2 = big[0].numel()
        off = [torch.empty(n2 + 1, dtype=torch.int64, device=dev) for _ in range(2)]
        ids = [torch.empty(n2 + 1024, dtype=torch.int32, device=dev) for _ in range(2)]
        for name, hint, k in (("default", ix.TOK_NARROW, 0), ("sweep", ix.TOK_NARROW | ix.TOK_SORTED, 1)):
            f = lambda s=False: ix.tokenize_device(big[0].data_ptr(), big[1].data_ptr(), big[2].data_ptr(), n2, off[k].data_ptr(),
                                                   ids[k].data_ptr(), ids[k].numel(), st, sync=s, hint=hint)
            h = f(True)
            dt = timed(f, 10)
            byts = 12 * n2 + 8 * (n2 + 1) + 4 * h + 12 * nu
            res[name + "_%dM" % (n2 // 1_000_000)] = {"us": round(dt * 1e6, 1), "frac": round(byts / dt / 8e12, 4), "hits": h}
        res["same_big"] = bool(torch.equal(off[0], off[1]) and torch.equal(ids[0][:h], ids[1][:h]))
        print(json.dumps(res), flush=True)
        del big, off, ids, batches, ix
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
