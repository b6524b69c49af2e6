#!/usr/bin/env python3
"""Per-phase shader-clock totals of k_tok_sweep (diagnostic build: sh tools/build_variant.sh tokstamps "-DGTARS_TOK_STAMPS=1" tokenize_lds.hip;
run with GTARS_AMD_LIB=build/variants/lib_tokstamps.so).  Wave 0 (look-back) and wave 1 of every workgroup."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import gtars_amd
from gtars_amd import synth, _lib

NAMES = ["query-load wait", "runs + window probes", "staging", "key search (LDS)", "interval walk (LDS)", "scan barrier", "look-back (wave 0) / stage ids",
         "barrier", "write (flush, offsets)", "closing barrier"]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
u = synth.make_universe(int(os.environ.get("NU", "100000")))
q = synth.make_queries(u, 1_000_000)
o = np.lexsort((q["start"], q["chrom"]))
q = {k: np.ascontiguousarray(v[o]) for k, v in q.items()}
ix = gtars_amd.OverlapIndex(u["chrom"], u["start"], u["end"], n_chrom=synth.N_CHROM)
fn = _lib.lib.gtars_debug_tok_stamps
fn.restype = C.c_int
buf = (C.c_ulonglong * 24)()
for n in [int(x) for x in os.environ.get("SIZES", "1000000,64000000").split(",")]:
    rep = max(n // 1_000_000, 1)
    big = [torch.from_numpy(q[k].view(np.int32)).to(dev).repeat(rep) for k in ("chrom", "start", "end")]
    if rep > 1:
        key = (big[0].to(torch.int64) & 0xFFFFFFFF) << 32 | (big[1].to(torch.int64) & 0xFFFFFFFF)
        order = torch.argsort(key, stable=True)
        big = [t[order].contiguous() for t in big]
        del key, order
    n = 1_000_000 * rep
    off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    ids = torch.empty(n + 1024, dtype=torch.int32, device=dev)
    f = lambda: ix.tokenize_device(big[0].data_ptr(), big[1].data_ptr(), big[2].data_ptr(), n, off.data_ptr(), ids.data_ptr(), ids.numel(), st,
                                   sync=False, hint=ix.TOK_NARROW | ix.TOK_SORTED)
    for _ in range(3): f()
    torch.cuda.synchronize()
    fn(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record()
    torch.cuda.synchronize()
    fn(buf, 0)
    v = list(buf)
    print(f"== {n} queries in order: {e0.elapsed_time(e1) * 1e3:.1f} us (diagnostic build)")
    for w in (0, 1):
        x = v[12 * w: 12 * w + 12]
        wg, tot = max(x[11], 1), sum(x[:10])
        print(f"  wave {w}: {wg} workgroups, {tot / wg:.0f} cycles each")
        for nme, y in zip(NAMES, x[:10]):
            print(f"     {nme:34s} {y / wg:10.0f}  {100 * y / max(tot, 1):5.1f} %")
    del big, off, ids
