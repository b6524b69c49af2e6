#!/usr/bin/env python3
"""Randomised differential soak: GPU tokenization / counts vs the oracle over many universe shapes.

Not part of the test suite (minutes of run time); run on the GPU box:  python tests/soak/fuzz_tokenize.py [rounds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import gtars_amd
import oracle

UNK = 0xFFFFFFFF


def one(seed):
    rng = np.random.default_rng(seed)
    n_chrom = int(rng.integers(1, 60))
    # (universes beyond the LDS key budget of the tokenizer kernel -- 130k regions -- up to 2M: units of 2 .. 16 blocks per key)
    n = int(rng.choice([1, 2, 3, 7, 50, 1000, 20_000, 150_000, 400_000, 2_000_000], p=[.12, .12, .12, .12, .12, .12, .12, .1, .04, .02]))
    span = int(rng.choice([50, 5_000, 1_000_000, 200_000_000, 4_000_000_000 // max(n_chrom, 1)]))
    if n >= 400_000:  # (a dense small span under millions of intervals is 1e10 ids per batch: keep the large ones genome-like)
        span = max(span, 200_000_000)
    wmax = int(rng.choice([1, 20, 2_000, max(2, span // 3)]))
    if n >= 400_000:
        wmax = min(wmax, 2_000)
    c = rng.integers(0, n_chrom, n)
    s = rng.integers(0, max(span, 2), n)
    w = rng.integers(0 if rng.random() < 0.3 else 1, wmax + 1, n)
    e = np.minimum(s.astype(np.int64) + w, 0xFFFFFFFF)
    if rng.random() < 0.2 and n > 3:          # a few chromosome-wide intervals
        k = rng.integers(0, n, 3)
        s[k] = 0
        e[k] = min(span + wmax, 0xFFFFFFFF)
    if rng.random() < 0.2 and n > 3:          # inverted universe intervals
        k = rng.integers(0, n, 3)
        e[k] = np.maximum(s[k].astype(np.int64) - 5, 0)
    if rng.random() < 0.3 and n > 1:          # a disjoint universe (ends ascend with the starts): wide queries take the run form
        order = np.lexsort((s, c))
        c, s, e = c[order], s[order], e[order].astype(np.int64)
        same = c[1:] == c[:-1]
        e[:-1] = np.where(same, np.minimum(e[:-1], s[1:]), e[:-1])
        e = np.maximum(e, s)
    val = rng.permutation(n).astype(np.uint32)
    if rng.random() < 0.5:                    # a sorted universe file: ids follow from the position (no id records)
        order = np.lexsort((e, s, c))
        c, s, e = c[order], s[order], e[order]
        val = np.arange(n, dtype=np.uint32)
    kind = gtars_amd.KIND_AILIST if rng.random() < 0.3 else gtars_amd.KIND_BITS
    os.environ["GTARS_TOK_ROUNDS"] = str(rng.choice(["0", "1", "2"]))    # launch geometry of the fused tokenizer
    os.environ["GTARS_TOK_GROUPS"] = str(rng.choice(["0", "1", "2"]))
    top_max = rng.choice(["", "64", "512"])
    if top_max:
        os.environ["GTARS_TOP_MAX"] = str(top_max)
    else:
        os.environ.pop("GTARS_TOP_MAX", None)
    buckets = rng.choice(["", "64", "4096", "16384"])  # bucket-table size of the LDS search (read at index build)
    if buckets:
        os.environ["GTARS_TOK_BUCKETS"] = str(buckets)
    else:
        os.environ.pop("GTARS_TOK_BUCKETS", None)
    for name, pr in (("GTARS_TOK_WIDE", 0.3), ("GTARS_TOK_NARROW", 0.15)):  # which build of the kernels a launch takes (capacity rule overridden)
        if rng.random() < pr:
            os.environ[name] = "1"
        else:
            os.environ.pop(name, None)
    # round 6: the sweep form of the tokenizer (k_tok_sweep) on a quarter of the cases, with random budgets (staged blocks per wave,
    # runs per wave and round, rounds per tile), and half of those cases with the batch in (chromosome, start) order
    sweep = rng.random() < 0.25
    for name, choices in (("GTARS_TOK_SWEEP", ["1"]), ("GTARS_TOK_SWEEP_BLOCKS", ["", "", "3", "40"]), ("GTARS_TOK_SWEEP_RUNS", ["", "", "0", "1"]),
                          ("GTARS_TOK_SWEEP_ROUNDS", ["", "1", "2"])):
        v = str(rng.choice(choices)) if sweep else ""
        if v:
            os.environ[name] = v
        else:
            os.environ.pop(name, None)
    gtars_amd.reload_env()  # (the library snapshots its switches at first use)
    g = gtars_amd.OverlapIndex(c, s, e, val, n_chrom=n_chrom, kind=kind)
    o = oracle.Index(c, s, e, val, n_chrom=n_chrom, kind=kind)
    nq = int(rng.choice([1, 5, 257, 4096, 4097, 70_001]))
    qc = rng.integers(0, n_chrom + 1, nq)
    qc = np.where(qc >= n_chrom, UNK, qc)
    qs = rng.integers(0, max(span, 2) + wmax, nq).astype(np.int64)
    qe = np.minimum(qs + rng.integers(0, max(2, wmax * 2), nq), 0xFFFFFFFF)
    if nq > 10:
        k = rng.integers(0, nq, max(1, nq // 50))
        qe[k] = np.maximum(qs[k] - rng.integers(0, 10, len(k)), 0)
        qs[:2] = 0xFFFFFFFF
        qe[:2] = 0xFFFFFFFF
    if (sweep and rng.random() < 0.5) or os.environ.get("SORTED") == "1":  # (SORTED=1: every case in order)
        order = np.lexsort((qs, qc))
        qc, qs, qe = qc[order], qs[order], qe[order]
    off_g, ids_g = g.tokenize(qc, qs, qe)
    off_o, ids_o = o.tokenize(qc, qs, qe)
    assert np.array_equal(off_g, off_o), ("offsets", seed)
    assert np.array_equal(ids_g, ids_o), ("ids", seed)
    for mo in (None, 1, 3):
        assert np.array_equal(g.count_overlaps(qc, qs, qe, mo), o.count_overlaps(qc, qs, qe, mo)), ("count", seed, mo)
    mo = [None, 2, 7][seed % 3]  # regions with payload: positions from the fused kernel + gather
    for a, b in zip(g.find_overlaps(qc, qs, qe, mo), o.find_overlaps_regions(qc, qs, qe, mo)):
        assert np.array_equal(a, b), ("find", seed, mo)
    if seed % 4 == 0:  # index-side subset: the bitmap-marking pass against the oracle's BTreeSet restatement
        mo = [None, 3][seed % 8 == 0]
        for a, b in zip(g.subset_by_overlaps(qc, qs, qe, mo), oracle.mco_subset_by_overlaps(o, qc, qs, qe, mo)):
            assert np.array_equal(a, b), ("subset", seed, mo)
    return n, nq, len(ids_o)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    base = int(sys.argv[2]) if len(sys.argv) > 2 else 1000  # first seed: another base = another set of shapes
    t = time.time()
    tot = 0
    for seed in range(rounds):
        n, nq, h = one(base + seed)
        tot += h
    print(f"fuzz_tokenize: {rounds} random configurations bit-exact vs the oracle ({tot} ids compared, {time.time() - t:.0f} s)")


if __name__ == "__main__":
    main()
