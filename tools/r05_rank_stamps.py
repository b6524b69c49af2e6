#!/usr/bin/env python3
"""Per-phase shader-clock totals of k_igd_sweep_rank (diagnostic build: tools/build_variant.sh igdstamps "-DIGD_STAMPS=1" igd_sweep.hip;
run with GTARS_AMD_LIB=build/variants/lib_igdstamps.so).  Wave 0 of every workgroup stamps s_memtime at the phase boundaries."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import gtars_amd
from gtars_amd import synth, _lib

NAMES = ["stage (loads issued, LDS writes, zero hist)", "barrier 1 wait", "queries", "barrier 2 wait", "scan + barrier 3", "bases + barrier 4",
         "final + end barrier"]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
F = 1000
db = synth.make_igd_db(int(os.environ.get("NDB", 50_000_000)), F)
g = gtars_amd.IgdIndex(db["chrom"], db["start"], db["end"], db["file"], n_chrom=synth.N_CHROM, n_files=F)
del db
hits = torch.zeros(F, dtype=torch.int64, device=dev)
fn = _lib.lib.gtars_debug_sweep_stamps
buf = (C.c_ulonglong * 8)()
q = synth.make_background_queries(int(os.environ.get("NQ", 10_000_000)))
d = [torch.from_numpy(np.ascontiguousarray(q[k]).view(np.int32)).to(dev) for k in ("chrom", "start", "end")]
n = d[0].numel()
for _ in range(2):
    g.count_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), n, hits.data_ptr(), 1, False, st)
torch.cuda.synchronize()
fn(buf, 1)
g.count_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), n, hits.data_ptr(), 1, False, st)
torch.cuda.synchronize()
fn(buf, 0)
v = list(buf)
wg = max(v[7], 1)
tot = sum(v[:7])
print(f"== pairwise: {wg} workgroups, {tot / wg:.0f} cycles per workgroup (s_memtime: 100 MHz ticks x ... see guide)")
for nme, x in zip(NAMES, v[:7]):
    print(f"   {nme:46s} {x / wg:10.0f}  {100 * x / tot:5.1f} %")
