#!/usr/bin/env python3
"""debug: the 'big skewed' soak case, stage by stage"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import gtars_amd, oracle
UNK = 0xFFFFFFFF
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
rng = np.random.default_rng(seed)
n_chrom = int(rng.integers(1, 5))
n = int(rng.choice([2_200_000, 3_000_000]))
F = int(rng.choice([1, 40, 3000]))
span = int(rng.choice([40_000_000, 200_000_000]))
wmax = int(rng.choice([300, 20_000]))
c = rng.integers(0, n_chrom, n); s = rng.integers(0, span, n); e = s + rng.integers(1, wmax, n); f = rng.integers(0, F, n)
g = gtars_amd.IgdIndex(c, s, e, f, np.arange(n), n_chrom=n_chrom, n_files=F)
o = oracle.Igd(); o.add_arrays(c, s, e, np.zeros(n, dtype=np.int64), f); o.n_files = F; o.finalize()
nq = int(rng.choice([1_050_000, 1_600_000]))
shape = rng.choice(["uniform", "skewed", "sorted"])
qc = rng.integers(0, n_chrom + 1, nq); qc = np.where(qc >= n_chrom, UNK, qc)
qs = rng.integers(0, span + wmax, nq).astype(np.int64)
if shape == "skewed":
    hot = rng.random(nq) < 0.99
    qc = np.where(hot, 0, qc); qs = np.where(hot, rng.integers(span // 3, span // 3 + 50_000, nq), qs)
qe = qs + rng.integers(1, max(2, wmax // 4), nq)
print(dict(n_chrom=n_chrom, n=n, F=F, span=span, wmax=wmax, nq=nq, shape=str(shape)), flush=True)
ref = o.count_set_overlaps(qc, qs, qe, 1, n_files=F)
refb = o.count_region_hits(qc, qs, qe, 1, n_files=F)
def run(tag):
    a = g.count_set_overlaps(qc, qs, qe, 1); b = g.count_region_hits(qc, qs, qe, 1)
    da = a.astype(np.int64) - ref.astype(np.int64); db = b.astype(np.int64) - refb.astype(np.int64)
    print(tag, "pair ok" if not da.any() else f"pair DIFF sum {da.sum()} nonzero {np.count_nonzero(da)} of {F} (ref total {ref.sum()})",
          "bin ok" if not db.any() else f"bin DIFF sum {db.sum()} nonzero {np.count_nonzero(db)}", flush=True)
run("default")
os.environ["GTARS_IGD_NO_FUSED_ROUTE"] = "1"; run("no fused route"); os.environ.pop("GTARS_IGD_NO_FUSED_ROUTE")
order = np.lexsort((qs, np.where(qc == UNK, n_chrom, qc)))
qc, qs, qe = qc[order], qs[order], qe[order]
run("sorted input")
# halves of the hot window
for frac in (0.5, 0.1, 0.02):
    m = int(nq * frac); qc2, qs2, qe2 = qc[:m], qs[:m], qe[:m]
    a = g.count_set_overlaps(qc2, qs2, qe2, 1); r = o.count_set_overlaps(qc2, qs2, qe2, 1, n_files=F)
    print("sorted prefix", m, "ok" if np.array_equal(a, r) else f"DIFF {int(a.sum()) - int(r.sum())}", flush=True)
