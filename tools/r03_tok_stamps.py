#!/usr/bin/env python3
"""Per-phase shader-clock totals of k_tok_lds (diagnostic build: tools/build_variant.sh tokstamps "-DGTARS_TOK_STAMPS=1"; run
with GTARS_AMD_LIB=build/variants/lib_tokstamps.so).  Wave 0 (look-back) and wave 1 of every workgroup."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import gtars_amd
from gtars_amd import synth, _lib

NAMES = ["query-load wait", "count (search + burst + masks)", "scan + barrier", "stage (wave 0: after the look-back)", "barrier", "write (flush, offsets)", "barrier", "look-back (wave 0) / stage"]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
u = synth.make_universe(100_000)
q = synth.make_queries(u, 1_000_000)
ix = gtars_amd.OverlapIndex(u["chrom"], u["start"], u["end"], n_chrom=synth.N_CHROM)
fn = _lib.lib.gtars_debug_tok_stamps
buf = (C.c_ulonglong * 24)()
for n in [int(x) for x in os.environ.get("SIZES", "1000000,64000000").split(",")]:
    rep = max(n // 1_000_000, 1)
    big = {k: torch.from_numpy(np.tile(q[k], rep).view(np.int32)).to(dev) for k in ("chrom", "start", "end")}
    n = 1_000_000 * rep
    off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    ids = torch.empty(n + 1024, dtype=torch.int32, device=dev)
    f = lambda: ix.tokenize_device(big["chrom"].data_ptr(), big["start"].data_ptr(), big["end"].data_ptr(), n, off.data_ptr(), ids.data_ptr(), ids.numel(), st, sync=False)
    for _ in range(3): f()
    torch.cuda.synchronize()
    fn(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record()
    torch.cuda.synchronize()
    fn(buf, 0)
    v = list(buf)
    print(f"== {n} queries: {e0.elapsed_time(e1) * 1e3:.1f} us (diagnostic build)")
    for w in (0, 1):
        x = v[12 * w: 12 * w + 12]
        wg, tot = max(x[11], 1), sum(x[:8])
        if w == 1: print(f"  look-back rounds: {x[8]} without a wait, {x[9]} after spinning (all workgroups)")
        print(f"  wave {w}: {wg} workgroups, {tot / wg:.0f} cycles each")
        for nme, y in zip(NAMES, x[:8]):
            print(f"     {nme:34s} {y / wg:10.0f}  {100 * y / tot:5.1f} %")
    del big, off, ids
