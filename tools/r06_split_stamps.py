#!/usr/bin/env python3
"""Per-phase shader-clock totals of k_split_pass (diagnostic build: tools/build_variant.sh spstamps "-DSP_STAMPS=1" sort.hip;
run with GTARS_AMD_LIB=build/variants/lib_spstamps.so).  Wave 0 of every workgroup stamps s_memtime at the phase boundaries of
the tile loop; printed: mean cycles per workgroup and share, for the two passes of one shuffled config-3 call."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import gtars_amd
from gtars_amd import synth, _lib

NAMES = ["prologue (coarse scan, column sums, fine offsets)", "wait for the tile's elements", "window + rank (LDS atomics)", "barrier",
         "layout scan + reservation issued", "reorder in LDS", "the reservation's answer", "barrier", "write-out", "closing barrier"]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
db = synth.make_igd_db(50_000_000, 1000)
g = gtars_amd.IgdIndex(db["chrom"], db["start"], db["end"], db["file"], n_chrom=synth.N_CHROM, n_files=1000)
del db
hits = torch.zeros(1000, dtype=torch.int64, device=dev)
lib = C.CDLL(_lib.lib._name)
buf = (C.c_ulonglong * 32)()
q = synth.make_background_queries(10_000_000)
d = [torch.from_numpy(np.ascontiguousarray(q[k]).view(np.int32)).to(dev) for k in ("chrom", "start", "end")]
n = d[0].numel()
for _ in range(2):
    g.count_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), n, hits.data_ptr(), 1, False, st)
torch.cuda.synchronize()
lib.gtars_debug_split_stamps(buf, 1)
g.count_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), n, hits.data_ptr(), 1, False, st)
torch.cuda.synchronize()
lib.gtars_debug_split_stamps(buf, 0)
v = list(buf)
for p, label in ((0, "pass A (coarse)"), (1, "pass B (fine)")):
    x = v[16 * p:16 * p + 16]
    wg = max(x[15], 1)
    tot = sum(x[:10])
    print(f"== {label}: {wg} workgroups, {tot / wg:.0f} cycles per workgroup")
    for nme, c in zip(NAMES, x[:10]):
        print(f"   {nme:50s} {c / wg:10.0f}  {100 * c / max(tot, 1):5.1f} %")
