#!/usr/bin/env python3
"""find_overlaps (regions with payload) on 1M queries vs the 100k universe: host call time and device kernel split.
GTARS_NO_LDS_PATH=1 shows the generic kernels."""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gtars_amd
from gtars_amd import synth, _lib
u = synth.make_universe(100_000); q = synth.make_queries(u, 1_000_000)
for kind in (gtars_amd.KIND_BITS, gtars_amd.KIND_AILIST):
    ix = gtars_amd.OverlapIndex(u["chrom"], u["start"], u["end"], n_chrom=synth.N_CHROM, kind=kind)
    ix.find_overlaps(q["chrom"], q["start"], q["end"])
    _lib.lib.gtars_prof_reset(); _lib.lib.gtars_prof_enable(1)
    t = time.perf_counter()
    for _ in range(5): ix.find_overlaps(q["chrom"], q["start"], q["end"])
    dt = (time.perf_counter() - t) / 5
    prof = _lib.prof_read(); _lib.lib.gtars_prof_enable(0)
    print(kind, "host call ms", round(dt * 1e3, 2), {k: round(v["total_ms"] / 5, 4) for k, v in prof.items()})
