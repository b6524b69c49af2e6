#!/usr/bin/env python3
"""Randomised differential soak: IGD batch counts (sweep and per-query kernels) vs the oracle's literal tile walk.

Run on the GPU box:  python tests/soak/fuzz_igd.py [rounds]      (small shapes, both kernels)
                     python tests/soak/fuzz_igd.py big [rounds]  (2-3M records, 1M+ queries: the two-level partition)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import gtars_amd
import oracle

UNK = 0xFFFFFFFF


def one(seed):
    rng = np.random.default_rng(seed)
    n_chrom = int(rng.integers(1, 6))
    n = int(rng.choice([1, 10, 3000, 20_000]))
    F = int(rng.choice([1, 3, 70, 3000]))
    span = int(rng.choice([100, 40_000, 3_000_000]))
    wmax = int(rng.choice([2, 300, 20_000, 70_000]))  # up to ~5 IGD tiles (nbp = 16384) per record
    c = rng.integers(0, n_chrom, n)
    s = rng.integers(-5 if rng.random() < 0.2 else 0, span, n)
    e = s + rng.integers(0 if rng.random() < 0.2 else 1, wmax, n)
    f = rng.integers(0, F, n)
    g = gtars_amd.IgdIndex(c, s, e, f, np.arange(n), n_chrom=n_chrom, n_files=F)
    o = oracle.Igd()
    o.add_arrays(c, s, e, np.arange(n), f)
    o.n_files = F
    o.finalize()
    nq = int(rng.choice([1, 33, 5000, 12_000]))
    qc = rng.integers(0, n_chrom + 1, nq)
    qc = np.where(qc >= n_chrom, UNK, qc)
    qs = rng.integers(0, span + wmax, nq).astype(np.int64)
    qe = qs + rng.integers(0, max(2, wmax), nq)
    if rng.random() < 0.5:
        order = np.lexsort((qs, np.where(qc == UNK, n_chrom, qc)))
        qc, qs, qe = qc[order], qs[order], qe[order]
    if nq > 40:
        qs[:5] = 0xFFFFFFF0       # negative as i32: clamped
        qe[5:10] = 0              # rejected
    if rng.random() < 0.7:
        os.environ["GTARS_IGD_SWEEP_MIN"] = "1"
    else:
        os.environ.pop("GTARS_IGD_SWEEP_MIN", None)
    # round 5: pairwise counts with min_overlap == 1 sweep in rank-histogram form; a quarter of the cases keep the candidate walk
    if rng.random() < 0.25:
        os.environ["GTARS_IGD_NO_RANK"] = "1"
    else:
        os.environ.pop("GTARS_IGD_NO_RANK", None)
    gtars_amd.reload_env()
    # min_overlap <= 0: the reference's tile walk also admits non-overlapping records (per-query kernels with the tile test)
    for mo in (1, int(rng.integers(2, 40)), int(rng.choice([0, -1, -30, -20_000]))):
        assert np.array_equal(g.count_set_overlaps(qc, qs, qe, mo), o.count_set_overlaps(qc, qs, qe, mo, n_files=F)), ("pair", seed, mo)
        assert np.array_equal(g.count_region_hits(qc, qs, qe, mo), o.count_region_hits(qc, qs, qe, mo, n_files=F)), ("bin", seed, mo)
    # several query sets in one call (gtars_igd_count_sets): the batch cut into 2-6 sets at random rows, some of them empty
    n_sets = int(rng.integers(2, 7))
    cuts = np.sort(rng.integers(0, nq + 1, n_sets - 1))
    if rng.random() < 0.3 and n_sets > 2:
        cuts[1] = cuts[0]  # an empty set
    bounds = [0] + cuts.tolist() + [nq]
    sets = [(qc[a:b] % 2**32, qs[a:b] % 2**32, qe[a:b] % 2**32) for a, b in zip(bounds[:-1], bounds[1:])]
    # round 5: sets that are each in (chromosome, start) order are swept as they arrive (no partition) -- half of the cases put every
    # set in the order the library checks for: valid queries by (chromosome, start as i32 clamped to 0), rejected ones behind them
    if rng.random() < 0.5:
        def in_order(t):
            c_, s_, e_ = (x.astype(np.int64) for x in t)
            si, ei = np.where(s_ >= 2**31, s_ - 2**32, s_), np.where(e_ >= 2**31, e_ - 2**32, e_)
            bad = (si >= ei) | (ei <= 0) | (c_ >= n_chrom)
            order = np.lexsort((np.where(bad, 0, np.maximum(si, 0)), np.where(bad, n_chrom, c_)))
            return tuple(x[order] for x in t)
        sets = [in_order(t) for t in sets]
    for binary in (True, False):
        mo = int(rng.choice([1, 1, 7]))
        got = g.count_sets(sets, mo, binary)
        ref = o.count_region_hits if binary else o.count_set_overlaps
        for k, t in enumerate(sets):
            assert np.array_equal(got[k], ref(t[0].astype(np.int64), t[1].astype(np.int64), t[2].astype(np.int64), mo, n_files=F)), ("sets", seed, binary, mo, k)
    if nq <= 5000 and n <= 3000:
        ok = (qs < 2**31) & (qe < 2**31)  # count_overlaps_per_query / find_overlaps_regionset take the query as it is
        mo = int(rng.choice([1, 5, 0, -30]))
        assert np.array_equal(g.count_overlaps_per_query(qc[ok], qs[ok], qe[ok], mo), o.count_overlaps_per_query(qc[ok], qs[ok], qe[ok], mo)), ("perq", seed, mo)
    return nq


def one_big(seed):
    """Databases of more than 1024 tiles with batches of 1M+ queries: the two-level, LDS-reordered partition
    (k_split_pass), its direct-slot path (skewed batches) and the device-side sorted / partition choice."""
    rng = np.random.default_rng(seed)
    n_chrom = int(rng.integers(1, 5))
    n = int(rng.choice([2_200_000, 3_000_000]))
    F = int(rng.choice([1, 40, 3000]))
    span = int(rng.choice([40_000_000, 200_000_000]))
    wmax = int(rng.choice([300, 20_000]))
    c = rng.integers(0, n_chrom, n)
    s = rng.integers(0, span, n)
    e = s + rng.integers(1, wmax, n)
    f = rng.integers(0, F, n)
    g = gtars_amd.IgdIndex(c, s, e, f, np.arange(n), n_chrom=n_chrom, n_files=F)
    o = oracle.Igd()
    o.add_arrays(c, s, e, np.zeros(n, dtype=np.int64), f)
    o.n_files = F
    o.finalize()
    nq = int(rng.choice([1_050_000, 1_600_000]))
    shape = rng.choice(["uniform", "skewed", "sorted"])
    qc = rng.integers(0, n_chrom + 1, nq)
    qc = np.where(qc >= n_chrom, UNK, qc)
    qs = rng.integers(0, span + wmax, nq).astype(np.int64)
    if shape == "skewed":  # almost everything in one window, a thin tail everywhere else
        hot = rng.random(nq) < 0.99
        qc = np.where(hot, 0, qc)
        qs = np.where(hot, rng.integers(span // 3, span // 3 + 50_000, nq), qs)
    qe = qs + rng.integers(1, max(2, wmax // 4), nq)
    if shape == "sorted":
        order = np.lexsort((qs, np.where(qc == UNK, n_chrom, qc)))
        qc, qs, qe = qc[order], qs[order], qe[order]
    os.environ.pop("GTARS_IGD_SWEEP_MIN", None)
    gtars_amd.reload_env()
    assert np.array_equal(g.count_set_overlaps(qc, qs, qe, 1), o.count_set_overlaps(qc, qs, qe, 1, n_files=F)), ("big pair", seed, shape)
    assert np.array_equal(g.count_region_hits(qc, qs, qe, 1), o.count_region_hits(qc, qs, qe, 1, n_files=F)), ("big bin", seed, shape)
    # the same batch as three sets sharing one pass (the two-level partition tags the pairs)
    b1, b2 = sorted(int(x) for x in rng.integers(0, nq + 1, 2))
    got = g.count_sets([(qc[:b1], qs[:b1], qe[:b1]), (qc[b1:b2], qs[b1:b2], qe[b1:b2]), (qc[b2:], qs[b2:], qe[b2:])], 1, True)
    for k, (a, b) in enumerate(((0, b1), (b1, b2), (b2, nq))):
        assert np.array_equal(got[k], o.count_region_hits(qc[a:b], qs[a:b], qe[a:b], 1, n_files=F)), ("big sets", seed, shape, k)
    return nq


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "big":
        t = time.time()
        base = int(sys.argv[3]) if len(sys.argv) > 3 else 9000  # first seed: another base = another set of shapes
        for seed in range(int(sys.argv[2])):
            one_big(base + seed)
            print(f"  big {seed + 1}, {time.time() - t:.0f} s", flush=True)
        print(f"fuzz_igd: {sys.argv[2]} large configurations bit-exact vs the oracle ({time.time() - t:.0f} s)")
        return
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    base = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    t = time.time()
    for seed in range(rounds):
        one(base + seed)
        if seed % 25 == 24:
            print(f"  {seed + 1} rounds, {time.time() - t:.0f} s", flush=True)
    print(f"fuzz_igd: {rounds} random configurations bit-exact vs the oracle ({time.time() - t:.0f} s)")


if __name__ == "__main__":
    main()
