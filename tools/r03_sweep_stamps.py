#!/usr/bin/env python3
"""Per-phase shader-clock totals of k_igd_sweep (diagnostic build: tools/build_variant.sh igdstamps "-DIGD_STAMPS=1";
run with GTARS_AMD_LIB=build/variants/lib_igdstamps.so).  Wave 0 of every workgroup stamps s_memtime at the phase boundaries;
printed: mean cycles per workgroup and share, for a pairwise and a binary config-3 call and the LOLA universe batch."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import gtars_amd
from gtars_amd import synth, _lib

NAMES = ["stage tile (loads + LDS writes + barrier)", "-", "-", "search (+loop head)", "pair loop", "barrier wait", "-"]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
F = int(os.environ.get("F", "1000"))
db = synth.make_igd_db(50_000_000, F)
g = gtars_amd.IgdIndex(db["chrom"], db["start"], db["end"], db["file"], n_chrom=synth.N_CHROM, n_files=F)
del db
hits = torch.zeros(F, dtype=torch.int64, device=dev)
fn = _lib.lib.gtars_debug_sweep_stamps
fr = _lib.lib.gtars_debug_route_stamps
buf = (C.c_ulonglong * 8)()
RNAMES = ["LDS fill", "routing loop (rest)", "barrier", "flush counters", "loop: wait for the columns", "loop: owner search", "loop: counters"]
def run(q, binary, label):
    d = [torch.from_numpy(np.ascontiguousarray(q[k]).view(np.int32)).to(dev) for k in ("chrom", "start", "end")]
    n = d[0].numel()
    for _ in range(2):
        g.count_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), n, hits.data_ptr(), 1, binary, st)
    torch.cuda.synchronize()
    fn(buf, 1)
    fr(buf, 1)
    g.count_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), n, hits.data_ptr(), 1, binary, st)
    torch.cuda.synchronize()
    fr(buf, 0)
    v = list(buf)
    if v[7]:
        tot = sum(v[:7])
        print(f"== {label}: k_igd_route, {v[7]} workgroups, {tot / v[7]:.0f} cycles per workgroup")
        for nme, x in zip(RNAMES, v[:7]):
            print(f"   {nme:28s} {x / v[7]:10.0f}  {100 * x / tot:5.1f} %")
    fn(buf, 0)
    v = list(buf)
    wg = max(v[7], 1)
    tot = sum(v[:7])
    print(f"== {label}: {wg} workgroups, {tot / wg:.0f} cycles per workgroup")
    for nme, x in zip(NAMES, v[:7]):
        print(f"   {nme:22s} {x / wg:10.0f}  {100 * x / tot:5.1f} %")
q = synth.make_background_queries(10_000_000)
run(q, False, "config 3 pairwise (410 queries per tile)")
run(q, True, "config 3 binary")
u = synth.make_universe(1_000_000, seed=3)
run(u, True, "LOLA universe, binary (41 queries per tile)")
