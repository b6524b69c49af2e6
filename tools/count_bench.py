#!/usr/bin/env python3
"""K2: count_overlaps on device-resident queries (count-only bytes: 12*Nq + 4*Nq + 12*Nu)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import gtars_amd
from gtars_amd import synth


def main():
    dev = torch.device("cuda:0")
    # OVERLAPPING=1: the ChIP-like C2' universe (overlaps kept, 1 % wide intervals: nested AIList sub-lists); KIND=ailist: the
    # reference's default IndexedRegionSet index (its order-free calls run on the flat companion's blocked structure)
    u = synth.make_universe(100_000, overlapping=os.environ.get("OVERLAPPING") == "1")
    base = synth.make_queries(u, 1_000_000)
    widen = int(os.environ.get("WIDEN", "0"))  # windowed counts: every query widened to this many bp
    if widen:
        base["end"] = (base["start"].astype(np.int64) + widen).clip(max=0x7FFFFFFF).astype(base["end"].dtype)
    kind = gtars_amd.KIND_AILIST if os.environ.get("KIND", "bits") == "ailist" else gtars_amd.KIND_BITS
    ix = gtars_amd.OverlapIndex(u["chrom"], u["start"], u["end"], n_chrom=synth.N_CHROM, kind=kind)
    st = torch.cuda.current_stream().cuda_stream
    for rep in (1, 64):
        qc, qs, qe = (torch.from_numpy(base[k].view(np.int32)).to(dev).repeat(rep) for k in ("chrom", "start", "end"))
        nq = qc.numel()
        counts = torch.empty(nq, dtype=torch.int32, device=dev)
        for path in (os.environ.get("LABEL", "k_count_lds (GTARS_NO_LDS_PATH=1 -> generic k_count)"),):
            f = lambda: ix.count_overlaps_device(qc.data_ptr(), qs.data_ptr(), qe.data_ptr(), nq, counts.data_ptr(), None, st)
            f(); torch.cuda.synchronize()
            reps = max(3, min(200, int(4e8 // nq)))
            t = time.perf_counter()
            for _ in range(reps):
                f()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / reps
            print(json.dumps({"path": path, "nq": nq, "us": round(dt * 1e6, 2), "gqps": round(nq / dt / 1e9, 2),
                              "hbm_frac": round((16 * nq + 12 * len(u["chrom"])) / dt / 8e12, 4), "sum": int(counts.sum())}))


if __name__ == "__main__":
    main()
