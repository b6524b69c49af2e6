"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit-exact.

Integer / index work: every comparison is exact equality (token ids AND their
order, CSR offsets, counts, per-file hit vectors).
"""
import os

import numpy as np
import pytest

import oracle
from oracle import KIND_AILIST, KIND_BITS

pytestmark = pytest.mark.gpu

BOTH = [KIND_BITS, KIND_AILIST]
UNK = 0xFFFFFFFF


@pytest.fixture(scope="module")
def ga():
    import gtars_amd

    assert gtars_amd.device_count() > 0, "no MI355X visible: -m gpu tests must run on the GPU box"
    return gtars_amd


def _pair(ga, chrom, start, end, val=None, n_chrom=None, kind=KIND_BITS):
    if n_chrom is None:
        n_chrom = int(max(chrom)) + 1 if len(chrom) else 0
    g = ga.OverlapIndex(chrom, start, end, val, n_chrom=n_chrom, kind=kind)
    o = oracle.Index(chrom, start, end, val, n_chrom=n_chrom, kind=kind)
    return g, o


def _assert_same_queries(g, o, qc, qs, qe, min_overlaps=(None,)):
    off_g, ids_g = g.tokenize(qc, qs, qe)
    off_o, ids_o = o.tokenize(qc, qs, qe)
    assert off_g.tolist() == off_o.tolist()
    assert ids_g.tolist() == ids_o.tolist()
    for mo in min_overlaps:
        assert g.count_overlaps(qc, qs, qe, mo).tolist() == o.count_overlaps(qc, qs, qe, mo).tolist()
        assert g.any_overlaps(qc, qs, qe, mo).tolist() == o.any_overlaps(qc, qs, qe, mo).tolist()
        fg = g.find_overlaps(qc, qs, qe, mo)
        fo = o.find_overlaps_regions(qc, qs, qe, mo)
        for a, b in zip(fg, fo):
            assert a.tolist() == b.tolist()


# ------------------------------------------------------------ reference KATs


@pytest.mark.parametrize("kind", BOTH)
def test_kat_abcd(ga, kind):
    # bits.rs:545-616 / ailist.rs:385-457
    g, o = _pair(ga, [0] * 4, [1, 3, 6, 8], [5, 7, 10, 12], kind=kind)
    _assert_same_queries(g, o, [0, 0, 0, 0, 9], [2, 9, 13, 0, 2], [4, 11, 15, 1, 4])
    off, ids = g.tokenize([0, 0, 0], [2, 9, 13], [4, 11, 15])
    assert sorted(ids[off[0]:off[1]].tolist()) == [0, 1]
    assert sorted(ids[off[1]:off[2]].tolist()) == [2, 3]
    assert off[3] == off[2]


def test_kat_ailist_26(ga):
    # ailist.rs:550-601
    from test_oracle_golden import AILIST_26

    s = [a for a, _ in AILIST_26]
    e = [b for _, b in AILIST_26]
    g, o = _pair(ga, [0] * 26, s, e, kind=KIND_AILIST)
    assert g.sublist_offsets(0) == [0, 24] == o.headers(0)
    off, fs, fe, _ = g.find_overlaps([0, 0, 0], [6, 30, 101], [8, 35, 150])
    assert off.tolist() == [0, 5, 8, 8]
    assert list(zip(fs[:5].tolist(), fe[:5].tolist())) == [(5, 15), (5, 15), (0, 10), (0, 10), (0, 30)]
    gs, ge, gv = g.stored(0)
    os_, oe, ov = o.stored(0)
    assert gs.tolist() == os_.tolist() and ge.tolist() == oe.tolist() and gv.tolist() == ov.tolist()


def test_kat_bits_order_and_maxlen(ga):
    g, o = _pair(ga, [0] * 5, [10, 10, 5, 10, 5], [30, 20, 50, 20, 50], [0, 1, 2, 3, 4])
    off, ids = g.tokenize([0], [0], [100])
    assert ids.tolist() == [2, 4, 1, 3, 0]
    assert g.max_len(0) == 45 == o.max_len(0)


@pytest.mark.parametrize("kind", BOTH)
def test_kat_mco(ga, kind):
    # multi_chrom_overlapper.rs:1070-1130, :878-943
    g, o = _pair(ga, [0, 0, 0], [150, 250, 500], [200, 350, 600], kind=kind)
    assert g.count_overlaps([0], [100], [300]).tolist() == [2]
    g, o = _pair(ga, [0], [150], [250], kind=kind)
    assert g.any_overlaps([0, 0], [100, 300], [200, 400]).tolist() == [True, False]
    g, o = _pair(ga, [0], [100], [110], kind=kind)
    assert g.count_overlaps([0], [105], [200], 5).tolist() == [1]
    assert g.count_overlaps([0], [105], [200], 6).tolist() == [0]
    assert g.any_overlaps([0], [105], [200], 6).tolist() == [False]
    g, o = _pair(ga, [0], [100], [200], kind=kind)
    assert g.count_overlaps([0, 99], [200, 100], [300, 200]).tolist() == [0, 0]


@pytest.mark.parametrize("kind", BOTH)
def test_kat_python_regionset_ops(ga, kind):
    # gtars-python/tests/test_regionset.py:37-54
    g, o = _pair(ga, [0, 0], [150, 550], [250, 650], kind=kind)
    qc, qs, qe = [0, 0, 0], [100, 300, 500], [200, 400, 600]
    assert g.count_overlaps(qc, qs, qe).tolist() == [1, 0, 1]
    assert g.any_overlaps(qc, qs, qe).tolist() == [True, False, True]
    off, idx = g.find_overlap_indices(qc, qs, qe)
    assert off.tolist() == [0, 1, 1, 2] and idx.tolist() == [0, 1]


@pytest.mark.parametrize("kind", BOTH)
def test_empty_index_and_empty_query(ga, kind):
    # multi_chrom_overlapper.rs:1132-1158, bits.rs:607-616
    g, o = _pair(ga, [], [], [], n_chrom=1, kind=kind)
    assert g.count_overlaps([0], [100], [200]).tolist() == [0]
    off, ids = g.tokenize([0], [100], [200])
    assert off.tolist() == [0, 0] and len(ids) == 0
    g, o = _pair(ga, [0], [100], [200], kind=kind)
    off, ids = g.tokenize([], [], [])
    assert off.tolist() == [0] and len(ids) == 0
    assert g.count_overlaps([], [], []).tolist() == []


def test_scoring_matrix_through_gpu(ga, golden_dir):
    # fragment_scoring.rs:178-206 (inverted end probe)
    import gzip
    import os

    cons = oracle.read_region_set(os.path.join(golden_dir, "consensus", "consensus1.bed"))
    ids = {}
    c = [ids.setdefault(r[0], len(ids)) for r in cons]
    g = ga.OverlapIndex(c, [r[1] for r in cons], [r[2] for r in cons], n_chrom=len(ids))
    mat = np.zeros((2, 4), dtype=np.int64)
    for row, name in enumerate(["fragments1.bed.gz", "fragments2.bed.gz"]):
        qc, qs, qe = [], [], []
        with gzip.open(os.path.join(golden_dir, "fragments", "region_scoring", name), "rt") as f:
            for line in f:
                p = line.split()
                if not p:
                    continue
                cid = ids.get(p[0], UNK)
                s, e = int(p[1]), int(p[2])
                qc += [cid, cid]
                qs += [s + 4, e - 5]
                qe += [s + 5, e - 6]
        _, hit = g.tokenize(qc, qs, qe)
        np.add.at(mat[row], hit, 1)
    assert mat.tolist() == [[2, 2, 1, 3], [4, 1, 3, 1]]


# ------------------------------------------------------- randomized differential


def _random_case(rng, n, nq, n_chrom, span, wmax, dup_frac=0.1, weird_frac=0.05):
    c = rng.integers(0, n_chrom, n)
    s = rng.integers(0, span, n)
    w = rng.integers(0, wmax, n)  # includes zero-length index intervals
    e = s + w
    ndup = int(n * dup_frac)
    if ndup and n:
        src = rng.integers(0, n, ndup)
        dst = rng.integers(0, n, ndup)
        c[dst], s[dst], e[dst] = c[src], s[src], e[src]
    qc = rng.integers(0, n_chrom + 2, nq)  # some unknown chromosome ids
    qc = np.where(qc >= n_chrom, UNK, qc)
    qs = rng.integers(0, span, nq)
    qw = rng.integers(0, wmax * 2, nq)
    qe = qs + qw
    nweird = int(nq * weird_frac)
    if nweird:
        k = rng.integers(0, nq, nweird)
        qe[k] = np.maximum(qs[k] - rng.integers(0, 5, nweird), 0)  # zero-length and inverted queries
    return c, s, e, qc, qs, qe


@pytest.mark.parametrize("kind", BOTH)
@pytest.mark.parametrize("seed,n,nq,n_chrom,span,wmax", [
    (1, 50, 300, 1, 200, 30),
    (2, 2000, 5000, 3, 20_000, 400),
    (3, 20_000, 30_000, 25, 1_000_000, 2_000),
    (4, 5000, 8000, 2, 5_000, 3_000),   # heavy coverage -> AIList decomposition into many sub-lists
    (5, 3000, 3000, 40, 100, 20),       # tiny span: many ties in (start,end)
])
def test_random_differential(ga, kind, seed, n, nq, n_chrom, span, wmax):
    rng = np.random.default_rng(seed)
    c, s, e, qc, qs, qe = _random_case(rng, n, nq, n_chrom, span, wmax)
    val = rng.permutation(n).astype(np.uint32)
    g, o = _pair(ga, c, s, e, val, n_chrom=n_chrom, kind=kind)
    for ch in range(n_chrom):
        for a, b in zip(g.stored(ch), o.stored(ch)):
            assert a.tolist() == b.tolist()
    _assert_same_queries(g, o, qc, qs, qe, min_overlaps=(None, 0, 1, 2, 5, 10))


def test_ailist_heavy_nesting(ga):
    # >= 10 of the next 19 intervals contained -> pushed to the next sub-list (ailist.rs:198-236)
    rng = np.random.default_rng(11)
    big_s = np.arange(0, 4000, 100)
    big_e = big_s + 5000
    small_s = rng.integers(0, 9000, 3000)
    small_e = small_s + rng.integers(1, 20, 3000)
    s = np.concatenate([big_s, small_s])
    e = np.concatenate([big_e, small_e])
    c = np.zeros(len(s), dtype=np.uint32)
    g, o = _pair(ga, c, s, e, kind=KIND_AILIST)
    assert len(o.headers(0)) >= 2
    assert g.sublist_offsets(0) == o.headers(0)
    qs = rng.integers(0, 9500, 4000)
    qe = qs + rng.integers(0, 300, 4000)
    _assert_same_queries(g, o, np.zeros(4000, dtype=np.uint32), qs, qe, min_overlaps=(None, 3))


@pytest.mark.parametrize("kind", BOTH)
def test_irs_find_overlap_indices_random(ga, kind):
    rng = np.random.default_rng(21)
    c, s, e, qc, qs, qe = _random_case(rng, 3000, 4000, 4, 8000, 300, dup_frac=0.3)
    g = ga.OverlapIndex(c, s, e, None, n_chrom=4, kind=kind)
    o = oracle.Index(c, s, e, None, n_chrom=4, kind=kind)
    for mo in (None, 4):
        og, ig = g.find_overlap_indices(qc, qs, qe, mo)
        oo, io = o.irs_find_overlaps(c, s, e, qc, qs, qe, mo)
        assert og.tolist() == oo.tolist()
        assert ig.tolist() == io.tolist()


@pytest.mark.parametrize("kind", BOTH)
def test_index_side_subset_kats_and_random(ga, kind, monkeypatch):
    """MultiChromOverlapper::subset_by_overlaps / intersect_all (multi_chrom_overlapper.rs:454-478, KAT :1044-1066) and
    IndexedRegionSet::intersect_all / subset_by_overlaps (indexed_region_set.rs:201-230, KAT :395-414) behind the C ABI:
    one bitmap-marking pass over the batch + compaction, against the oracle's BTreeSet restatement."""
    g, o = _pair(ga, [0, 0, 1], [100, 300, 500], [200, 400, 600], n_chrom=2, kind=kind)
    q = ([0, 1], [150, 550], [250, 650])
    assert [x.tolist() for x in g.subset_by_overlaps(*q)] == [[0, 1], [100, 500], [200, 600]]
    assert g.subset_source_indices(*q).tolist() == [0, 2]
    assert [len(x) for x in g.subset_by_overlaps([], [], [])] == [0, 0, 0] and len(g.subset_source_indices([], [], [])) == 0
    e0 = ga.OverlapIndex([], [], [], None, n_chrom=1, kind=kind)
    assert len(e0.subset_by_overlaps([0], [100], [200])[0]) == 0 and len(e0.subset_source_indices([0], [100], [200])) == 0
    rng = np.random.default_rng(77)
    for n, nq, span, dup in ((3000, 4000, 8000, 0.3), (40_000, 30_000, 3_000_000, 0.0), (500, 200_000, 2_000_000, 0.1)):
        c, s, e, qc, qs, qe = _random_case(rng, n, nq, 4, span, 300, dup_frac=dup)
        g = ga.OverlapIndex(c, s, e, None, n_chrom=4, kind=kind)
        o = oracle.Index(c, s, e, None, n_chrom=4, kind=kind)
        for forced_generic in (False, True):
            if forced_generic:
                monkeypatch.setenv("GTARS_NO_LDS_PATH_FOR_TEST", "1")
            for mo in (None, 1, 4, 40):
                got = g.subset_by_overlaps(qc, qs, qe, mo)
                exp = oracle.mco_subset_by_overlaps(o, qc, qs, qe, mo)
                assert all(a.tolist() == b.tolist() for a, b in zip(got, exp)), (n, mo)
                assert g.subset_source_indices(qc, qs, qe, mo).tolist() == oracle.irs_subset_by_overlaps(o, c, s, e, qc, qs, qe, mo).tolist()
        monkeypatch.delenv("GTARS_NO_LDS_PATH_FOR_TEST")
    # the device form: bit p <=> stored position p is hit
    import torch

    dev = torch.device("cuda:0")
    d = [torch.from_numpy(np.ascontiguousarray(x, dtype=np.uint32).view(np.int32)).to(dev) for x in (qc, qs, qe)]
    if kind == KIND_BITS:
        mark = torch.full(((len(c) + 31) // 32,), -1, dtype=torch.int32, device=dev)  # the call zeroes it
        g.mark_overlapped_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), len(qc), mark.data_ptr(), None,
                                 torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        bits = np.unpackbits(mark.cpu().numpy().view(np.uint8), bitorder="little")[: len(c)].astype(bool)
        got = []
        base = 0
        for ch in range(4):
            ss, ee, _ = g.stored(ch)
            sel = bits[base : base + len(ss)]
            got += [(ch, int(a), int(b)) for a, b in zip(ss[sel], ee[sel])]
            base += len(ss)
        exp = oracle.mco_subset_by_overlaps(o, qc, qs, qe)
        assert sorted(set(got)) == list(zip(*[x.tolist() for x in exp]))


def test_nested_ailist_index_answers_order_free_calls_on_its_flat_companion(ga):
    """The reference's default IndexedRegionSet index is AIList (indexed_region_set.rs:111-113), and count / any / find_overlaps
    (sorted unique source rows) / subset_by_overlaps / intersect_all do not depend on its enumeration order: an AIList-kind
    index with NESTED sub-lists (ChIP-like C2' universe: 1 % of the intervals 5-100 kbp wide) answers them on the blocked
    structure of a flat companion -- the LDS kernels run (profiling facts), the results are the oracle's, and the device bitmap
    of gtars_mark_overlapped_device is in THIS index's stored (sub-list major) order.  AIList-order enumeration stays exact."""
    import torch

    from gtars_amd import _lib

    rng = np.random.default_rng(2024)
    n, n_chrom, span = 60_000, 3, 12_000_000  # an interval per 600 bp; 2 % of them 5-100 kbp wide: they contain dozens of others
    c = rng.integers(0, n_chrom, n).astype(np.uint32)
    s = rng.integers(0, span, n).astype(np.uint32)
    w = np.where(rng.random(n) < 0.02, rng.integers(5_000, 100_000, n), rng.integers(100, 900, n))
    e = (s + w).astype(np.uint32)
    nq = 300_000
    qc = rng.integers(0, n_chrom + 1, nq).astype(np.uint32)
    qc[qc == n_chrom] = UNK
    qs = rng.integers(0, span + 50_000, nq).astype(np.uint32)
    qe = (qs + rng.integers(0, 700, nq)).astype(np.uint32)
    g, o = _pair(ga, c, s, e, n_chrom=n_chrom, kind=KIND_AILIST)
    assert max(len(g.sublist_offsets(ch)) for ch in range(n_chrom)) > 1  # nested: no blocked structure of its own
    _lib.lib.gtars_prof_reset()
    _lib.lib.gtars_prof_enable(1)
    try:
        for mo in (None, 1, 30):
            assert np.array_equal(g.count_overlaps(qc, qs, qe, mo), o.count_overlaps(qc, qs, qe, mo).astype(np.uint32)), mo
            assert np.array_equal(g.any_overlaps(qc, qs, qe, mo), o.any_overlaps(qc, qs, qe, mo)), mo
        n_small = 40_000
        og, ig = g.find_overlap_indices(qc[:n_small], qs[:n_small], qe[:n_small])
        oo, io = o.irs_find_overlaps(c, s, e, qc[:n_small], qs[:n_small], qe[:n_small])
        assert og.tolist() == oo.tolist() and ig.tolist() == io.tolist()
        for mo in (None, 25):
            got = g.subset_by_overlaps(qc, qs, qe, mo)
            exp = oracle.mco_subset_by_overlaps(o, qc, qs, qe, mo)
            assert all(a.tolist() == b.tolist() for a, b in zip(got, exp)), mo
            assert g.subset_source_indices(qc, qs, qe, mo).tolist() == oracle.irs_subset_by_overlaps(o, c, s, e, qc, qs, qe, mo).tolist()
        dev = torch.device("cuda:0")
        d = [torch.from_numpy(np.ascontiguousarray(x, dtype=np.uint32).view(np.int32)).to(dev) for x in (qc, qs, qe)]
        mark = torch.full(((len(c) + 31) // 32,), -1, dtype=torch.int32, device=dev)  # the call zeroes it
        g.mark_overlapped_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), len(qc), mark.data_ptr(), None,
                                 torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        prof = _lib.prof_read()
    finally:
        _lib.lib.gtars_prof_enable(0)
    assert "k_count_lds" in prof and "k_mark_lds" in prof and "k_count" not in prof, sorted(prof)
    bits = np.unpackbits(mark.cpu().numpy().view(np.uint8), bitorder="little")[: len(c)].astype(bool)
    got, base = [], 0
    for ch in range(n_chrom):
        ss, ee, _ = g.stored(ch)  # this index's stored order: sub-list major
        sel = bits[base : base + len(ss)]
        got += [(ch, int(a), int(b)) for a, b in zip(ss[sel], ee[sel])]
        base += len(ss)
    exp = oracle.mco_subset_by_overlaps(o, qc, qs, qe)
    assert sorted(set(got)) == list(zip(*[x.tolist() for x in exp]))
    # enumeration in AIList::find order (ailist.rs:153-178) is still this index's own business
    off_g, ids_g = g.tokenize(qc[:n_small], qs[:n_small], qe[:n_small])
    off_o, ids_o = o.tokenize(qc[:n_small], qs[:n_small], qe[:n_small])
    assert np.array_equal(off_g, off_o) and np.array_equal(ids_g, ids_o)


def test_nested_ailist_enumeration_order_from_the_flat_companion(ga, monkeypatch):
    """Round 5: a tokenizer_type = "ailist" universe with NESTED sub-lists (config.rs:27-41; any overlapping universe) no longer
    enumerates on the one-thread-per-query kernel: the hit set comes from the flat companion's LDS tokenizer and
    k_ailist_reorder puts every query's hits into AIList::find order -- sub-list after sub-list, each from its last candidate
    down (ailist.rs:153-178, 238-263).  Offsets and ids == the oracle's, on heavily nested data (up to dozens of hits per query,
    several sub-list levels, duplicates of the same interval with different values, queries with > 32 hits: the heap-sort
    branch), through the host call, the device call, a too-small id buffer + fill, and with a min-overlap filter through
    find_overlaps; the generic kernel (GTARS_AILIST_NO_REORDER) agrees.  Which path ran is a profiling fact."""
    import torch

    monkeypatch.setenv("GTARS_AILIST_REORDER_MAX_DEPTH", "1000")  # (this universe is deep: by default it would keep the generic kernel)
    _lib = ga._lib
    rng = np.random.default_rng(77)
    n, n_chrom, span = 50_000, 3, 6_000_000
    c = rng.integers(0, n_chrom, n).astype(np.uint32)
    s = rng.integers(0, span, n).astype(np.uint32)
    w = np.where(rng.random(n) < 0.03, rng.integers(5_000, 300_000, n), rng.integers(100, 900, n))
    e = (s + w).astype(np.uint32)
    s[100:140] = s[100]  # forty copies of one interval (values differ: their order among themselves is the stored order)
    e[100:140] = e[100]
    c[100:140] = c[100]
    v = rng.permutation(n).astype(np.uint32)
    g = ga.OverlapIndex(c, s, e, v, n_chrom=n_chrom, kind=KIND_AILIST)
    o = oracle.Index(c, s, e, v, n_chrom=n_chrom, kind=KIND_AILIST)
    assert max(len(g.sublist_offsets(ch)) for ch in range(n_chrom)) > 2  # several levels
    nq = 120_000
    qc = rng.integers(0, n_chrom + 1, nq).astype(np.uint32)
    qc[qc == n_chrom] = UNK
    qs = rng.integers(0, span + 10_000, nq).astype(np.uint32)
    qe = (qs + np.where(rng.random(nq) < 0.01, rng.integers(20_000, 60_000, nq), rng.integers(0, 900, nq))).astype(np.uint32)
    qs[:50] = s[100]  # queries on the forty copies
    qe[:50] = e[100]
    qc[:50] = c[100]
    off_o, ids_o = o.tokenize(qc, qs, qe)
    assert int(np.diff(off_o.astype(np.int64)).max()) > 32
    _lib.lib.gtars_prof_reset()
    _lib.lib.gtars_prof_enable(1)
    off_g, ids_g = g.tokenize(qc, qs, qe)
    facts = _lib.prof_read()
    _lib.lib.gtars_prof_enable(0)
    assert "ailist_nested_on_lds" in facts and "k_tok_lds" in facts and "k_ailist_reorder" in facts, sorted(facts)
    assert np.array_equal(off_g, off_o) and np.array_equal(ids_g, ids_o)
    # device call into a too-small buffer: CAPACITY, offsets complete; the fill pass (generic kernel, AIList order) completes the ids
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    d = [torch.from_numpy(x.view(np.int32)).to(dev) for x in (qc, qs, qe)]
    off = torch.empty(nq + 1, dtype=torch.int64, device=dev)
    small = torch.empty(len(ids_o) // 3, dtype=torch.int32, device=dev)
    with pytest.raises(ga.CapacityError):
        g.tokenize_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, off.data_ptr(), small.data_ptr(), small.numel(), st)
    assert np.array_equal(off.cpu().numpy().view(np.uint64), off_o)
    full = torch.empty(len(ids_o), dtype=torch.int32, device=dev)
    g.fill_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, off.data_ptr(), full.data_ptr(), st)
    torch.cuda.synchronize()
    assert np.array_equal(full.cpu().numpy().view(np.uint32), ids_o)
    h = g.tokenize_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, off.data_ptr(), full.data_ptr(), full.numel(), st)
    assert h == len(ids_o) and np.array_equal(full.cpu().numpy().view(np.uint32), ids_o)
    monkeypatch.setenv("GTARS_AILIST_NO_REORDER", "1")
    off_x, ids_x = g.tokenize(qc, qs, qe)
    assert np.array_equal(off_x, off_o) and np.array_equal(ids_x, ids_o)
    monkeypatch.delenv("GTARS_AILIST_NO_REORDER")
    # the default depth line: this universe (mean depth ~14 intervals per covered position) keeps the generic kernel
    monkeypatch.delenv("GTARS_AILIST_REORDER_MAX_DEPTH")
    _lib.lib.gtars_prof_reset()
    _lib.lib.gtars_prof_enable(1)
    off_y, ids_y = g.tokenize(qc, qs, qe)
    facts = _lib.prof_read()
    _lib.lib.gtars_prof_enable(0)
    assert "ailist_nested_on_lds" not in facts and np.array_equal(off_y, off_o) and np.array_equal(ids_y, ids_o)


def test_config1_1k_by_1k(ga):
    # BASELINE config 1: 1k x 1k single chromosome
    from gtars_amd import synth

    c, s, e = synth.make_single_chrom(1000, 1)
    qc, qs, qe = synth.make_single_chrom(1000, 2)
    for kind in BOTH:
        g, o = _pair(ga, c, s, e, n_chrom=1, kind=kind)
        _assert_same_queries(g, o, qc, qs, qe, min_overlaps=(None, 50))
        og, ig = g.find_overlap_indices(qc, qs, qe)
        oo, io = o.irs_find_overlaps(c, s, e, qc, qs, qe)
        assert og.tolist() == oo.tolist() and ig.tolist() == io.tolist()


@pytest.mark.parametrize("overlapping", [False, True])
def test_config2_tokenize_1m_vs_100k(ga, overlapping):
    # BASELINE config 2: 1M hg38-shaped queries vs a 100k-region universe, bit-exact ids and offsets
    from gtars_amd import synth

    u = synth.make_universe(100_000, overlapping=overlapping)
    q = synth.make_queries(u, 1_000_000)
    for kind in ([KIND_BITS] if not overlapping else BOTH):
        g, o = _pair(ga, u["chrom"], u["start"], u["end"], n_chrom=synth.N_CHROM, kind=kind)
        off_g, ids_g = g.tokenize(q["chrom"], q["start"], q["end"])
        off_o, ids_o = o.tokenize(q["chrom"], q["start"], q["end"])
        assert np.array_equal(off_g, off_o)
        assert np.array_equal(ids_g, ids_o)
        assert np.array_equal(g.count_overlaps(q["chrom"], q["start"], q["end"]),
                              o.count_overlaps(q["chrom"], q["start"], q["end"]).astype(np.uint32))


def test_device_pointer_path_and_capacity(ga):
    """gtars_tokenize_device on torch buffers; CAPACITY overflow -> offsets valid, fill pass completes."""
    import torch
    from gtars_amd import synth

    u = synth.make_universe(20_000)
    q = synth.make_queries(u, 200_001)  # not a multiple of the tile, exercises the ragged tail
    g, o = _pair(ga, u["chrom"], u["start"], u["end"], n_chrom=synth.N_CHROM)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a.astype(np.int64)).to(dev).to(torch.int32)  # u32 bit patterns in int32
    qc = torch.from_numpy(q["chrom"].view(np.int32)).to(dev)
    qs = torch.from_numpy(q["start"].view(np.int32)).to(dev)
    qe = torch.from_numpy(q["end"].view(np.int32)).to(dev)
    nq = len(q["chrom"])
    offsets = torch.empty(nq + 1, dtype=torch.int64, device=dev)
    ids = torch.empty(2 * nq, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    h = g.tokenize_device(qc.data_ptr(), qs.data_ptr(), qe.data_ptr(), nq, offsets.data_ptr(), ids.data_ptr(),
                          ids.numel(), stream)
    off_o, ids_o = o.tokenize(q["chrom"], q["start"], q["end"])
    assert h == len(ids_o)
    assert np.array_equal(offsets.cpu().numpy().view(np.uint64), off_o)
    assert np.array_equal(ids[:h].cpu().numpy().view(np.uint32), ids_o)
    # misaligned views (scalar load path) give the same answer
    qc1, qs1, qe1 = qc[1:], qs[1:], qe[1:]
    h1 = g.tokenize_device(qc1.data_ptr(), qs1.data_ptr(), qe1.data_ptr(), nq - 1, offsets.data_ptr(),
                           ids.data_ptr(), ids.numel(), stream)
    off_1, ids_1 = o.tokenize(q["chrom"][1:], q["start"][1:], q["end"][1:])
    assert h1 == len(ids_1)
    assert np.array_equal(ids[:h1].cpu().numpy().view(np.uint32), ids_1)
    # too-small capacity: status CAPACITY, offsets still complete; the fill pass then produces the ids
    small = torch.empty(16, dtype=torch.int32, device=dev)
    with pytest.raises(ga.CapacityError):
        g.tokenize_device(qc.data_ptr(), qs.data_ptr(), qe.data_ptr(), nq, offsets.data_ptr(), small.data_ptr(),
                          small.numel(), stream)
    assert np.array_equal(offsets.cpu().numpy().view(np.uint64), off_o)
    g.fill_device(qc.data_ptr(), qs.data_ptr(), qe.data_ptr(), nq, offsets.data_ptr(), ids.data_ptr(), stream)
    torch.cuda.synchronize()
    assert np.array_equal(ids[: len(ids_o)].cpu().numpy().view(np.uint32), ids_o)
    assert np.array_equal(offsets.cpu().numpy().view(np.uint64), off_o)  # (the caller's offsets are read-only to the fill)


def test_fill_device_both_ways_and_on_wide_queries(ga, monkeypatch):
    """gtars_fill_device after a GTARS_ERR_CAPACITY: through the fused tokenizer (an index the LDS kernels serve) and through the
    generic fill kernel (GTARS_NO_LDS_PATH_FOR_TEST), narrow and hit-heavy batches, both index kinds."""
    import torch

    rng = np.random.default_rng(99)
    n_chrom, span = 4, 2_500_000
    C_, S, E = _disjoint_universe(rng, n_chrom, 5_000, span)
    dev = torch.device("cuda:0")
    for kind in BOTH:
        g, o = _pair(ga, C_, S, E, n_chrom=n_chrom, kind=kind)
        for typical in (300, 40_000):
            qc, qs, qe = _wide_queries(rng, n_chrom, span, 20_000, typical=typical)
            qc = np.where(qc >= n_chrom, 1, qc)
            off_o, ids_o = o.tokenize(qc, qs, qe)
            d = [torch.from_numpy(np.ascontiguousarray(x, dtype=np.uint32).view(np.int32)).to(dev) for x in (qc, qs, qe)]
            for no_lds in (False, True):
                if no_lds:
                    monkeypatch.setenv("GTARS_NO_LDS_PATH_FOR_TEST", "1")
                off = torch.zeros(len(qc) + 1, dtype=torch.int64, device=dev)
                h = g.tokenize_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), len(qc), off.data_ptr(), 0, 0)
                assert h == len(ids_o)
                ids = torch.full((h + 64,), -3, dtype=torch.int32, device=dev)
                g.fill_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), len(qc), off.data_ptr(), ids.data_ptr())
                torch.cuda.synchronize()
                got = ids.cpu().numpy()
                assert np.array_equal(got[:h].view(np.uint32), ids_o), (kind, typical, no_lds)
                assert (got[h:] == -3).all()
                assert np.array_equal(off.cpu().numpy().view(np.uint64), off_o)
                if no_lds:
                    monkeypatch.delenv("GTARS_NO_LDS_PATH_FOR_TEST")


def test_tokenizer_build_is_the_callers_choice(ga):
    """Round 5: which build of the fused tokenizer runs is a HINT the caller can give (gtars_tokenize_device_ex), not only a
    function of its id buffer's size.  A config-2-like batch into a 4x over-allocated buffer runs the narrow build when told so
    (by the capacity rule alone it would run the wide one), a two-pass caller's fill (gtars_fill_device_n: total hits known)
    runs the narrow build for ~1 id per query and the wide one for a hit-heavy batch and never writes past the total; every
    combination gives the oracle's offsets and ids.  Which build ran is read from the library's profiling facts."""
    import torch

    from gtars_amd import synth

    _lib = ga._lib
    u = synth.make_universe(20_000)
    g = ga.OverlapIndex(u["chrom"], u["start"], u["end"], n_chrom=synth.N_CHROM)
    o = oracle.Index(u["chrom"], u["start"], u["end"], None, n_chrom=synth.N_CHROM)
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream

    def facts_of(fn):
        _lib.lib.gtars_prof_reset()
        _lib.lib.gtars_prof_enable(1)
        r = fn()
        torch.cuda.synchronize()
        f = _lib.prof_read()
        _lib.lib.gtars_prof_enable(0)
        return r, {k for k in f if k.startswith("tok_build")}

    q = synth.make_queries(u, 200_000)
    wide_q = dict(q)
    wide_q["end"] = (q["start"].astype(np.int64) + 2_000_000).clip(max=2**31 - 1).astype(np.uint32)  # ~13 ids per query
    for batch, heavy in ((q, False), (wide_q, True)):
        off_o, ids_o = o.tokenize(batch["chrom"], batch["start"], batch["end"])
        d = [torch.from_numpy(batch[k].view(np.int32)).to(dev) for k in ("chrom", "start", "end")]
        nq = len(batch["chrom"])
        off = torch.empty(nq + 1, dtype=torch.int64, device=dev)
        big = torch.empty(max(4 * nq, len(ids_o)) + 64, dtype=torch.int32, device=dev)  # over-allocated: room for 4 ids per query
        run = lambda hint: g.tokenize_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, off.data_ptr(), big.data_ptr(),
                                             big.numel(), st, hint=hint)
        for hint, want in ((g.TOK_AUTO, "tok_build_wide"), (g.TOK_NARROW, "tok_build_narrow"), (g.TOK_WIDE, "tok_build_wide")):
            h, facts = facts_of(lambda: run(hint))
            assert facts == {want}, (heavy, hint, facts)
            assert h == len(ids_o) and np.array_equal(off.cpu().numpy().view(np.uint64), off_o)
            assert np.array_equal(big[:h].cpu().numpy().view(np.uint32), ids_o)
        # the two-pass flow: offsets only, then the fill with the total the caller read back
        h = g.tokenize_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, off.data_ptr(), 0, 0, st)
        ids = torch.full((h + 32,), -5, dtype=torch.int32, device=dev)
        _, facts = facts_of(lambda: g.fill_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, off.data_ptr(), ids.data_ptr(), st,
                                                  total_hits=h))
        assert facts == {"tok_build_wide" if heavy else "tok_build_narrow"}, (heavy, facts)
        got = ids.cpu().numpy()
        assert np.array_equal(got[:h].view(np.uint32), ids_o) and (got[h:] == -5).all()
        # a total that is too small bounds the writes (the contract's violation does not run off the buffer)
        ids.fill_(-5)
        g.fill_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, off.data_ptr(), ids.data_ptr(), st, total_hits=h // 2)
        torch.cuda.synchronize()
        got = ids.cpu().numpy()
        assert np.array_equal(got[: h // 2].view(np.uint32), ids_o[: h // 2]) and (got[h // 2:] == -5).all()
    with pytest.raises(ValueError):
        g.tokenize_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, off.data_ptr(), 0, 0, st, hint=7)


@pytest.mark.parametrize("rep", [16, 256])
def test_full_size_properties(ga, rep):
    """Size-independent properties at the scaling sizes of SURVEY section 8d (1.6e7 and 2.56e8 queries), which
    the oracle is not asked to check in full: offsets monotone, offsets[-1] == H, counts == diff(offsets),
    ids within the universe, and -- the batch being `rep` copies of a 1M base -- every copy tokenizes
    exactly as the oracle tokenizes the base."""
    import torch
    from gtars_amd import synth

    u = synth.make_universe(100_000)
    base = synth.make_queries(u, 1_000_000)
    g, o = _pair(ga, u["chrom"], u["start"], u["end"], n_chrom=synth.N_CHROM)
    dev = torch.device("cuda:0")
    qc, qs, qe = (torch.from_numpy(base[k].view(np.int32)).to(dev).repeat(rep) for k in ("chrom", "start", "end"))
    nq = qc.numel()
    off_o, ids_o = o.tokenize(base["chrom"], base["start"], base["end"])
    offsets = torch.empty(nq + 1, dtype=torch.int64, device=dev)
    ids = torch.empty(rep * len(ids_o) + 64, dtype=torch.int32, device=dev)
    counts = torch.empty(nq, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    h = g.tokenize_device(qc.data_ptr(), qs.data_ptr(), qe.data_ptr(), nq, offsets.data_ptr(), ids.data_ptr(),
                          ids.numel(), st)
    g.count_overlaps_device(qc.data_ptr(), qs.data_ptr(), qe.data_ptr(), nq, counts.data_ptr(), None, st)
    torch.cuda.synchronize()
    assert h == rep * len(ids_o)
    d = offsets[1:] - offsets[:-1]
    assert int(offsets[0]) == 0 and int(offsets[-1]) == h
    assert bool((d >= 0).all())
    assert bool((d == counts.to(torch.int64)).all())
    del d, counts
    assert int(ids[:h].max()) < len(u["chrom"]) and int(ids[:h].min()) >= 0
    # periodicity, checked on the device for every copy: ids and per-query counts repeat with period 1M
    ids_base = torch.from_numpy(ids_o.view(np.int32)).to(dev)
    assert bool((ids[:h].view(rep, len(ids_o)) == ids_base).all())
    off_base = torch.from_numpy(off_o.astype(np.int64)).to(dev)
    per_copy = (offsets[:-1].view(rep, -1) - offsets[:-1].view(rep, -1)[:, :1])
    assert bool((per_copy == off_base[:-1]).all())


# ---------------------------------------------------------------------- IGD


def _igd_pair(ga, c, s, e, f, v=None, n_chrom=None, n_files=None):
    g = ga.IgdIndex(c, s, e, f, v, n_chrom=n_chrom, n_files=n_files)
    o = oracle.Igd()
    o.add_arrays(c, s, e, v if v is not None else np.zeros(len(c), dtype=np.int64), f)
    o.n_files = n_files if n_files is not None else o.n_files
    o.finalize()
    return g, o


def test_igd_kats(ga):
    # igd.rs:914-1016, 1161-1221
    g, o = _igd_pair(ga, [0, 0, 0], [100, 300, 150], [200, 400, 250], [0, 0, 1], n_files=2)
    assert g.count_set_overlaps([0], [120], [180]).tolist() == [1, 1]
    assert g.count_set_overlaps([0], [350], [380]).tolist() == [1, 0]
    assert g.count_set_overlaps([0], [500], [600]).tolist() == [0, 0]
    g, o = _igd_pair(ga, [0], [100], [200], [0], n_files=1)
    assert [int(g.count_set_overlaps([0], [190], [250], mo)[0]) for mo in (1, 10, 11)] == [1, 1, 0]
    g, o = _igd_pair(ga, [0], [10000], [20000], [0], n_files=1)
    assert g.total_records() == 2 == o.total_records()
    assert g.count_set_overlaps([0, 0, 0], [11000, 17000, 15000], [12000, 18000, 19000]).tolist() == [3]
    g, o = _igd_pair(ga, [0, 0, 0], [100, 120, 140], [200, 220, 240], [0, 0, 0], n_files=1)
    assert g.count_set_overlaps([0], [150], [190]).tolist() == [3]
    assert g.count_region_hits([0], [150], [190]).tolist() == [1]
    # unknown chrom, invalid queries (igd.rs:514-517)
    assert g.count_set_overlaps([7, 0, 0], [150, 190, 150], [190, 150, 150]).tolist() == [0]


def test_igd_two_set_kats(ga):
    # igd.rs:1256-1369
    g, o = _igd_pair(ga, [0, 0, 0], [100, 300, 500], [200, 400, 600], [0, 0, 0], [0, 1, 2], n_files=1)
    q, s = g.find_overlaps_regionset([0, 0, 0], [150, 550, 700], [350, 650, 800])
    assert sorted(zip(q.tolist(), s.tolist())) == [(0, 0), (0, 1), (1, 2)]
    g, o = _igd_pair(ga, [0, 0, 0], [100, 150, 500], [200, 250, 600], [0, 0, 0], [0, 1, 2], n_files=1)
    assert g.count_overlaps_per_query([0, 0, 0], [160, 550, 700], [180, 580, 800]).tolist() == [2, 1, 0]
    g, o = _igd_pair(ga, [0], [10000], [40000], [0], [0], n_files=1)
    q, s = g.find_overlaps_regionset([0], [15000], [35000])
    assert list(zip(q.tolist(), s.tolist())) == [(0, 0)]


@pytest.mark.parametrize("seed,n,nq,F,span,wmax", [
    (1, 4000, 3000, 5, 200_000, 40_000),     # long records spanning several 16384-bp tiles
    (2, 20_000, 20_000, 300, 2_000_000, 900),
    (3, 3000, 2000, 9000, 100_000, 500),     # F > LDS bins -> global-atomic path
])
def test_igd_random_differential(ga, seed, n, nq, F, span, wmax):
    rng = np.random.default_rng(seed)
    n_chrom = 3
    c = rng.integers(0, n_chrom, n)
    s = rng.integers(0, span, n)
    e = s + rng.integers(0, wmax, n)  # some zero-length records are dropped by add()
    s[:5] = -3  # negative starts dropped
    f = rng.integers(0, F, n)
    v = np.arange(n)
    g, o = _igd_pair(ga, c, s, e, f, v, n_chrom=n_chrom, n_files=F)
    assert len(g) == int(((s >= 0) & (e > s)).sum())
    assert g.total_records() == o.total_records()
    qc = rng.integers(0, n_chrom + 1, nq)
    qc = np.where(qc >= n_chrom, UNK, qc)
    qs = rng.integers(0, span, nq)
    qe = qs + rng.integers(0, wmax, nq)
    # tile-edge cases
    qs[:50] = (rng.integers(1, 10, 50) * 16384)
    qe[:50] = qs[:50] + rng.integers(1, 40000, 50)
    for mo in (1, 2, 10):
        assert g.count_set_overlaps(qc, qs, qe, mo).tolist() == o.count_set_overlaps(qc, qs, qe, mo, n_files=F).tolist()
        assert g.count_region_hits(qc, qs, qe, mo).tolist() == o.count_region_hits(qc, qs, qe, mo, n_files=F).tolist()
    if n <= 4000:
        for mo in (1, 10):
            assert g.count_overlaps_per_query(qc, qs, qe, mo).tolist() == o.count_overlaps_per_query(qc, qs, qe, mo).tolist()
            gq, gs = g.find_overlaps_regionset(qc, qs, qe, mo)
            oq, os_ = o.find_overlaps_regionset(qc, qs, qe, mo)
            assert gq.tolist() == oq.tolist() and gs.tolist() == os_.tolist()  # same walk order


def test_lola_contingency_device(ga):
    import ctypes as C

    import torch
    from gtars_amd._lib import check, lib

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    uh = rng.integers(0, 1000, 2000)
    vh = uh + rng.integers(-5, 5000, 2000).clip(min=-3)
    vh = np.maximum(vh, 0)
    a, b, c, d = oracle.lola_contingency(uh, vh, 1000, 100000)
    tu, tv = torch.from_numpy(uh.astype(np.int64)).to(dev), torch.from_numpy(vh.astype(np.int64)).to(dev)
    out = [torch.empty(2000, dtype=torch.int64, device=dev) for _ in range(4)]
    check(lib.gtars_lola_contingency_device(tu.data_ptr(), tv.data_ptr(), 2000, 1000, 100000,
                                            *[o.data_ptr() for o in out], torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    for got, exp in zip(out, (a, b, c, d)):
        assert got.cpu().numpy().tolist() == exp.tolist()


# ------------------------------------------------------------------ K1: device radix sort


@pytest.mark.parametrize("n,n_chrom,span", [(5000, 3, 300), (70_000, 25, 1_000_000), (40_000, 700, 5_000)])
def test_device_sort_builds_the_same_index(ga, monkeypatch, n, n_chrom, span):
    """The device radix sort (stable LSD passes) must give exactly the reference's stable orders:
    Bits (start,end,input) per chromosome, IGD (start, insertion); many ties on purpose."""
    rng = np.random.default_rng(n)
    c = rng.integers(0, n_chrom, n)
    s = rng.integers(0, span, n)
    e = s + rng.integers(0, 50, n)
    val = rng.permutation(n).astype(np.uint32)
    o = oracle.Index(c, s, e, val, n_chrom=n_chrom, kind=KIND_BITS)
    monkeypatch.setenv("GTARS_DEVICE_SORT", "1")
    g_dev = ga.OverlapIndex(c, s, e, val, n_chrom=n_chrom, kind=KIND_BITS)
    monkeypatch.setenv("GTARS_DEVICE_SORT", "0")
    g_host = ga.OverlapIndex(c, s, e, val, n_chrom=n_chrom, kind=KIND_BITS)
    for ch in range(0, n_chrom, max(1, n_chrom // 40)):
        exp = [x.tolist() for x in o.stored(ch)]
        assert [x.tolist() for x in g_dev.stored(ch)] == exp
        assert [x.tolist() for x in g_host.stored(ch)] == exp
        assert g_dev.max_len(ch) == o.max_len(ch)
    qc = rng.integers(0, n_chrom, 5000)
    qs = rng.integers(0, span, 5000)
    qe = qs + rng.integers(0, 100, 5000)
    off_o, ids_o = o.tokenize(qc, qs, qe)
    off_g, ids_g = g_dev.tokenize(qc, qs, qe)
    assert off_g.tolist() == off_o.tolist() and ids_g.tolist() == ids_o.tolist()
    # IGD: walk order of find_overlaps_regionset depends on the (start, insertion) order
    F = 11
    f = rng.integers(0, F, n)
    monkeypatch.setenv("GTARS_DEVICE_SORT", "1")
    gi, oi = _igd_pair(ga, c, s, e + 1, f, np.arange(n), n_chrom=n_chrom, n_files=F)
    assert gi.count_set_overlaps(qc, qs, qe + 1).tolist() == oi.count_set_overlaps(qc, qs, qe + 1, 1, n_files=F).tolist()
    q2 = slice(0, 400)
    gq, gs = gi.find_overlaps_regionset(qc[q2], qs[q2], qe[q2] + 1)
    oq, os_ = oi.find_overlaps_regionset(qc[q2], qs[q2], qe[q2] + 1)
    assert gq.tolist() == oq.tolist() and gs.tolist() == os_.tolist()


@pytest.mark.parametrize("top_max", ["64", "300", "4096"])
def test_lds_path_with_sampled_top_level(ga, monkeypatch, top_max):
    """Universes too large for the LDS top level use a sampled top (top_shift > 0) plus a short search of
    the block keys in L2; force that path with a tiny LDS budget and compare with the oracle."""
    from gtars_amd import synth

    monkeypatch.setenv("GTARS_TOP_MAX", top_max)
    rng = np.random.default_rng(int(top_max))
    for overlapping in (False, True):
        u = synth.make_universe(60_000, seed=11, overlapping=overlapping)
        q = synth.make_queries(u, 150_001, seed=12)
        g, o = _pair(ga, u["chrom"], u["start"], u["end"], n_chrom=synth.N_CHROM)
        off_g, ids_g = g.tokenize(q["chrom"], q["start"], q["end"])
        off_o, ids_o = o.tokenize(q["chrom"], q["start"], q["end"])
        assert np.array_equal(off_g, off_o) and np.array_equal(ids_g, ids_o)
    # dense, heavily overlapping index with long intervals: multi-block tails, many hits per query
    n = 30_000
    c = rng.integers(0, 3, n)
    s = rng.integers(0, 200_000, n)
    e = s + rng.integers(1, 5_000, n)
    g, o = _pair(ga, c, s, e, n_chrom=3)
    qc = rng.integers(0, 4, 20_000)
    qc = np.where(qc >= 3, UNK, qc)
    qs = rng.integers(0, 210_000, 20_000)
    qe = qs + rng.integers(0, 3_000, 20_000)
    off_g, ids_g = g.tokenize(qc, qs, qe)
    off_o, ids_o = o.tokenize(qc, qs, qe)
    assert np.array_equal(off_g, off_o) and np.array_equal(ids_g, ids_o)
    assert len(ids_o) > 50 * len(qc)  # really many hits per query


@pytest.mark.parametrize("seed,n,nq,F,span,wmax", [
    (1, 4000, 3000, 5, 200_000, 40_000),       # records much longer than a 2048-record tile spans
    (2, 60_000, 50_000, 300, 2_000_000, 900),  # many tiles per chromosome
    (3, 9000, 4000, 9000, 100_000, 500),       # more files than LDS-friendly, still < 16384 bins
    (4, 30_000, 20_000, 40, 50_000, 3_000),    # dense: most queries straddle tile boundaries
])
def test_igd_sweep_matches_oracle(ga, monkeypatch, seed, n, nq, F, span, wmax):
    """The batch sweep (sorted queries x database tiles in LDS) against the oracle's literal tile walk,
    pairwise and binary, including invalid / clamped / unknown-chromosome queries."""
    monkeypatch.setenv("GTARS_IGD_SWEEP_MIN", "1")
    rng = np.random.default_rng(100 + seed)
    n_chrom = 3
    c = rng.integers(0, n_chrom, n)
    s = rng.integers(0, span, n)
    e = s + rng.integers(1, wmax, n)
    f = rng.integers(0, F, n)
    g, o = _igd_pair(ga, c, s, e, f, np.arange(n), n_chrom=n_chrom, n_files=F)
    qc = rng.integers(0, n_chrom + 1, nq)
    qc = np.where(qc >= n_chrom, UNK, qc)
    qs = rng.integers(0, span, nq).astype(np.int64)
    qe = qs + rng.integers(0, wmax, nq)           # some zero-length (invalid) queries
    qs[:20] = 0xFFFFFFF0                           # negative as i32: clamped to 0
    qe[:20] = rng.integers(1, span, 20)
    qe[20:40] = 0                                  # end <= 0: rejected
    for mo in (1, 7):
        assert g.count_set_overlaps(qc, qs, qe, mo).tolist() == o.count_set_overlaps(qc, qs, qe, mo, n_files=F).tolist()
        assert g.count_region_hits(qc, qs, qe, mo).tolist() == o.count_region_hits(qc, qs, qe, mo, n_files=F).tolist()
    # a shuffled batch is grouped by owner tile in one partition pass (no radix sort), binary counts with
    # min_overlap == 1 go through the per-record pme_file instead of a list of credited files
    _lib = ga._lib
    _lib.lib.gtars_prof_reset()
    _lib.lib.gtars_prof_enable(1)
    exp_b = o.count_region_hits(qc, qs, qe, 1, n_files=F).tolist()
    assert g.count_region_hits(qc, qs, qe, 1).tolist() == exp_b
    names = set(_lib.prof_read())
    _lib.lib.gtars_prof_enable(0)
    assert "k_ms_scatter" in names and "k_radix_scatter" not in names, names
    # ... and the alternatives agree: full radix sort of the batch, credited-file list, per-query kernel
    monkeypatch.setenv("GTARS_IGD_FULL_SORT", "1")
    assert g.count_region_hits(qc, qs, qe, 1).tolist() == exp_b
    assert g.count_set_overlaps(qc, qs, qe, 1).tolist() == o.count_set_overlaps(qc, qs, qe, 1, n_files=F).tolist()
    monkeypatch.delenv("GTARS_IGD_FULL_SORT")
    monkeypatch.setenv("GTARS_IGD_NO_PME", "1")
    assert g.count_region_hits(qc, qs, qe, 1).tolist() == exp_b
    monkeypatch.delenv("GTARS_IGD_NO_PME")
    monkeypatch.setenv("GTARS_NO_IGD_SWEEP", "1")
    assert g.count_region_hits(qc, qs, qe, 1).tolist() == exp_b


def test_igd_routing_chunks_of_whole_steps_on_a_long_shuffled_batch(ga, monkeypatch):
    """Round 6: for long batches the routing kernel's chunks are whole 4096-query steps (fewer workgroups than CUs, no partial
    last step) and the first step's columns are requested in front of the table copy.  8.4M shuffled queries -- 256 chunks of
    32 816 become 228 of 36 864, the last one short and ragged -- against a two-level partition (1270 tiles), pairwise and binary,
    checked against the oracle and against the plain ceil(n / workgroups) geometry."""
    rng = np.random.default_rng(77)
    n, F, n_chrom, span = 2_600_000, 40, 3, 120_000_000
    c = rng.integers(0, n_chrom, n)
    s = rng.integers(0, span, n)
    e = s + rng.integers(1, 400, n)
    f = rng.integers(0, F, n)
    g, o = _igd_pair(ga, c, s, e, f, n_chrom=n_chrom, n_files=F)
    nq = 8 * 4096 * 256 + 3 * 4096 + 7
    qc = rng.integers(0, n_chrom + 1, nq)
    qc = np.where(qc >= n_chrom, UNK, qc)
    qs = rng.integers(0, span, nq).astype(np.int64)
    qe = qs + rng.integers(0, 300, nq)
    qs[-3:] = 0xFFFFFFF0  # (the ragged tail: clamped starts)
    qe[-3:] = 5000
    want_p = o.count_set_overlaps(qc, qs, qe, 1, n_files=F).tolist()
    want_b = o.count_region_hits(qc, qs, qe, 1, n_files=F).tolist()
    _lib = ga._lib
    _lib.lib.gtars_prof_reset()
    _lib.lib.gtars_prof_enable(1)
    assert g.count_set_overlaps(qc, qs, qe, 1).tolist() == want_p
    names = set(_lib.prof_read())
    _lib.lib.gtars_prof_enable(0)
    assert "k_split_pass" in names, names  # the two-level partition (the whole-step geometry only applies there)
    assert g.count_region_hits(qc, qs, qe, 1).tolist() == want_b
    monkeypatch.setenv("GTARS_IGD_ROUTE_NO_WHOLE_STEPS", "1")
    ga.reload_env()
    try:
        assert g.count_set_overlaps(qc, qs, qe, 1).tolist() == want_p
    finally:
        monkeypatch.delenv("GTARS_IGD_ROUTE_NO_WHOLE_STEPS")
        ga.reload_env()


def test_igd_non_positive_min_overlap_follows_the_tile_walk(ga):
    """igd.rs:772-846 with min_overlap <= 0: the walk also admits records that do NOT overlap the query, depending on the
    16384-bp tile they fall in.  Every IGD query against the oracle's literal tile walk, incl. records that straddle tile
    borders, queries spanning several tiles, queries past the contig's last tile and database values that repeat."""
    rng = np.random.default_rng(2024)
    n, nq, F, n_chrom = 30_000, 20_000, 40, 3
    c = rng.integers(0, n_chrom, n)
    s = rng.integers(0, 400_000, n)
    s[: n // 4] = (rng.integers(1, 24, n // 4) * 16384 - rng.integers(0, 300, n // 4)).clip(0)  # right before tile borders
    e = s + rng.integers(1, 2_000, n)
    e[:200] = s[:200] + rng.integers(16_000, 70_000, 200)  # records that span several tiles
    f = rng.integers(0, F, n)
    for values in (np.arange(n), rng.integers(0, n // 3, n)):
        g, o = _igd_pair(ga, c, s, e, f, values, n_chrom=n_chrom, n_files=F)
        qc = rng.integers(0, n_chrom + 1, nq)
        qc[qc == n_chrom] = UNK
        qs = rng.integers(0, 430_000, nq).astype(np.int64)
        qs[: nq // 5] = (rng.integers(1, 26, nq // 5) * 16384 + rng.integers(-200, 200, nq // 5)).clip(0)
        qe = qs + rng.integers(1, 600, nq)
        qe[:300] = qs[:300] + rng.integers(16_384, 60_000, 300)  # queries over several tiles
        for mo in (0, -1, -50, -1000, -40_000):
            assert g.count_set_overlaps(qc, qs, qe, mo).tolist() == o.count_set_overlaps(qc, qs, qe, mo, n_files=F).tolist(), mo
            assert g.count_region_hits(qc, qs, qe, mo).tolist() == o.count_region_hits(qc, qs, qe, mo, n_files=F).tolist(), mo
        for mo in (0, -50, -1000):
            sub = slice(0, 4000)
            assert g.count_overlaps_per_query(qc[sub], qs[sub], qe[sub], mo).tolist() == \
                o.count_overlaps_per_query(qc[sub], qs[sub], qe[sub], mo).tolist(), mo
            gq, gs_ = g.find_overlaps_regionset(qc[sub], qs[sub], qe[sub], mo)
            oq, os_ = o.find_overlaps_regionset(qc[sub], qs[sub], qe[sub], mo)
            assert gq.tolist() == oq.tolist() and gs_.tolist() == os_.tolist(), mo


def test_igd_sweep_presorted_batch_skips_the_sort(ga, monkeypatch):
    """A batch already in (chromosome, start) order takes the sweep without the partition; same vectors as the oracle
    and as the forced-partition path.  The choice is made on the device (no host round trip); what the tests observe
    is the device-side flag itself (profiling mode reads it back: `igd_batch_in_owner_order` / `igd_batch_partitioned`),
    never a kernel time."""
    monkeypatch.setenv("GTARS_IGD_SWEEP_MIN", "1")
    rng = np.random.default_rng(321)
    n, nq, F, n_chrom, span = 40_000, 3_000_000, 64, 4, 500_000
    c = rng.integers(0, n_chrom, n)
    s = rng.integers(0, span, n)
    e = s + rng.integers(1, 2_000, n)
    f = rng.integers(0, F, n)
    g, o = _igd_pair(ga, c, s, e, f, np.arange(n), n_chrom=n_chrom, n_files=F)
    qc = rng.integers(0, n_chrom, nq)
    qs = rng.integers(0, span, nq).astype(np.int64)
    order = np.lexsort((qs, qc))
    qc, qs = qc[order], qs[order]
    qe = qs + rng.integers(1, 300, nq)
    qc = np.concatenate([qc, np.full(50, UNK)])  # unknown chromosomes sort last
    qs = np.concatenate([qs, rng.integers(0, span, 50)])
    qe = np.concatenate([qe, qs[-50:] + 10])
    exp_p = o.count_set_overlaps(qc, qs, qe, 1, n_files=F).tolist()
    exp_b = o.count_region_hits(qc, qs, qe, 1, n_files=F).tolist()
    _lib = ga._lib

    def profiled():
        _lib.lib.gtars_prof_reset()
        _lib.lib.gtars_prof_enable(1)
        assert g.count_set_overlaps(qc, qs, qe, 1).tolist() == exp_p
        assert g.count_region_hits(qc, qs, qe, 1).tolist() == exp_b
        prof = _lib.prof_read()
        _lib.lib.gtars_prof_enable(0)
        return prof

    def took(prof):
        return {k: v["launches"] for k, v in prof.items() if k.startswith("igd_batch_")}

    g.count_region_hits(qc[:1000], qs[:1000], qe[:1000], 1)  # builds the database's pme_file outside the profiled calls
    in_order = profiled()
    assert any(k.startswith("k_igd_sweep") for k in in_order) and "k_gather2_u32" not in in_order, in_order
    assert took(in_order) == {"igd_batch_in_owner_order": 2}, in_order
    monkeypatch.setenv("GTARS_IGD_ALWAYS_SORT", "1")
    forced = profiled()
    assert took(forced) == {"igd_batch_partitioned": 2}, forced
    # and a shuffled batch partitions by itself
    monkeypatch.delenv("GTARS_IGD_ALWAYS_SORT")
    sh = rng.permutation(len(qc))
    qc, qs, qe = qc[sh], qs[sh], qe[sh]
    shuffled = profiled()
    assert took(shuffled) == {"igd_batch_partitioned": 2}, shuffled


def test_bits_count_matches_reference_formula(ga):
    """Bits::count (bits.rs:337-344): equal to the oracle's restatement bit for bit -- including the wrapping
    results for zero-length / inverted queries -- and equal to find().len() for ordinary queries."""
    rng = np.random.default_rng(77)
    n, nq, n_chrom = 30_000, 20_000, 5
    c = rng.integers(0, n_chrom - 1, n)          # the last chromosome stays empty
    s = rng.integers(0, 1_000_000, n)
    e = s + rng.integers(1, 3_000, n)
    g, o = _pair(ga, c, s, e, n_chrom=n_chrom, kind=KIND_BITS)
    qc = rng.integers(0, n_chrom + 1, nq)
    qc = np.where(qc >= n_chrom, UNK, qc)
    qs = rng.integers(0, 1_000_000, nq)
    qe = qs + rng.integers(1, 5_000, nq)
    k = rng.integers(0, nq, 500)
    qe[k] = np.maximum(qs[k] - rng.integers(0, 50, 500), 0)   # zero-length and inverted
    qs[:5] = 0xFFFFFFFF                                          # start + 1 wraps
    got = g.bits_count(qc, qs, qe)
    exp = np.array([o.bits_count(int(a) if a != UNK else 0xFFFFFFFF, int(b), int(d)) for a, b, d in zip(qc, qs, qe)],
                   dtype=np.uint64)
    assert np.array_equal(got, exp)
    ordinary = (qe > qs) & (qs != 0xFFFFFFFF)
    assert np.array_equal(got[ordinary], g.count_overlaps(qc, qs, qe)[ordinary].astype(np.uint64))
    # the reference KAT (bits.rs tests / tests/test_oracle_golden.py)
    g2, o2 = _pair(ga, np.zeros(3, dtype=np.uint32), [1, 5, 10], [6, 9, 15], n_chrom=1, kind=KIND_BITS)
    assert g2.bits_count([0], [5], [11]).tolist() == [o2.bits_count(0, 5, 11)]
    with pytest.raises(Exception):
        ga.OverlapIndex(c, s, e, n_chrom=n_chrom, kind=KIND_AILIST).bits_count(qc[:4], qs[:4], qe[:4])


def test_tokenize_wide_nested_and_degenerate_universe_intervals(ga):
    """The LDS kernel starts a scan at the first block whose prefix-max END is > q_start instead of
    lower_bound(q_start - max_len): same hits, same order -- with chromosome-wide intervals, deep nesting,
    zero-length and inverted universe intervals, and queries at the extremes of u32."""
    rng = np.random.default_rng(2024)
    n_chrom, span = 4, 3_000_000
    s = rng.integers(0, span, 60_000)
    e = s + rng.integers(1, 400, 60_000)
    c = rng.integers(0, n_chrom, 60_000)
    # a few intervals that cover most of a chromosome (max_len ~ span), early and late in start order
    ws = np.array([0, 10, 500_000, 2_900_000, 1_000, 2_999_000])
    we = np.array([span, span - 5, 2_500_000, 2_999_999, 2_000_000, 0xFFFFFFFF])
    wc = np.array([0, 0, 1, 1, 2, 2])
    # zero-length and inverted universe intervals (never validated by the reference)
    zs = rng.integers(0, span, 300)
    ze = np.where(rng.random(300) < 0.5, zs, np.maximum(zs.astype(np.int64) - rng.integers(1, 50, 300), 0))
    zc = rng.integers(0, n_chrom, 300)
    S = np.concatenate([s, ws, zs]); E = np.concatenate([e, we, ze]); C_ = np.concatenate([c, wc, zc])
    val = rng.permutation(len(S)).astype(np.uint32)
    g, o = _pair(ga, C_, S, E, val, n_chrom=n_chrom, kind=KIND_BITS)
    nq = 50_000
    qc = rng.integers(0, n_chrom, nq)
    qs = rng.integers(0, span, nq).astype(np.int64)
    qe = qs + rng.integers(0, 2_000, nq)
    qs[:50] = 0xFFFFFFFF; qe[:50] = 0xFFFFFFFF          # nothing has end > 0xFFFFFFFF
    qs[50:100] = 0; qe[50:100] = 0xFFFFFFFF             # everything on the chromosome, in order
    qs[100:150] = span - 1; qe[100:150] = span + 10
    k = rng.integers(150, nq, 400)
    qe[k] = np.maximum(qs[k] - rng.integers(0, 30, 400), 0)   # zero-length / inverted queries
    off_g, ids_g = g.tokenize(qc, qs, qe)
    off_o, ids_o = o.tokenize(qc, qs, qe)
    assert np.array_equal(off_g, off_o)
    assert np.array_equal(ids_g, ids_o)
    assert len(ids_o) > 20 * nq  # the wide intervals make every query a multi-hit query


def _disjoint_universe(rng, n_chrom, per_chrom, span):
    """sorted, disjoint (ends ascend with the starts), with touching neighbours, equal ends, zero-length intervals and chromosomes
    of odd / tiny sizes"""
    C_, S, E = [], [], []
    for c in range(n_chrom):
        n = per_chrom if c else 1  # chromosome 0: a single interval
        n += c % 2
        cuts = np.sort(rng.choice(span, 2 * n, replace=False))
        s, e = cuts[0::2].copy(), cuts[1::2].copy()
        k = rng.integers(0, n - 1, max(n // 50, 1)) if n > 1 else []
        for i in k:
            e[i] = s[i + 1]            # touching
        for i in (rng.integers(0, n - 1, max(n // 80, 1)) if n > 1 else []):
            e[i] = e[i + 1]            # equal ends (the earlier one contains nothing: starts still ascend)
        for i in rng.integers(0, n, max(n // 100, 1)):
            e[i] = s[i] if i == 0 or e[i - 1] <= s[i] else e[i]  # zero-length
        e = np.maximum.accumulate(np.maximum(e, s))              # (keep the ends ascending after the edits)
        C_.append(np.full(n, c)); S.append(s); E.append(e)
    return np.concatenate(C_), np.concatenate(S), np.concatenate(E)


def _wide_queries(rng, n_chrom, span, nq, typical):
    qc = rng.integers(0, n_chrom + 1, nq)  # (n_chrom: an unknown chromosome)
    qs = rng.integers(0, span, nq).astype(np.int64)
    w = np.where(rng.random(nq) < 0.4, rng.integers(0, 400, nq), rng.integers(0, 2 * typical, nq))
    qe = qs + w
    k = rng.integers(0, nq, nq // 20)
    qe[k] = span + rng.integers(0, 1000, len(k))       # to the chromosome's end and past it
    k = rng.integers(0, nq, nq // 50)
    qs[k] = 0; qe[k] = 0xFFFFFFFF                      # everything on the chromosome
    k = rng.integers(0, nq, nq // 50)
    qe[k] = np.maximum(qs[k] - rng.integers(0, 30, len(k)), 0)  # zero-length / inverted
    return qc, qs, qe


@pytest.mark.parametrize("kind", BOTH)
@pytest.mark.parametrize("explicit_ids", [False, True])
def test_wide_queries_on_disjoint_universes_take_the_run_form(ga, monkeypatch, kind, explicit_ids):
    """Hit-heavy batches on a universe whose ends ascend with the starts: a wide query's tail is measured by a second search
    (tail_run) and, with position-derived ids, its ids leave by wave-wide stores -- against the oracle's scan, and against the
    walked form (GTARS_TOK_NO_RUNS); one and two rounds / wave groups per tile; tokenize, count, any, find (position view)."""
    rng = np.random.default_rng(77 + int(explicit_ids))
    n_chrom, span = 5, 4_000_000
    C_, S, E = _disjoint_universe(rng, n_chrom, 9_001, span)
    val = rng.permutation(len(S)).astype(np.uint32) if explicit_ids else None
    g, o = _pair(ga, C_, S, E, val, n_chrom=n_chrom, kind=kind)
    qc, qs, qe = _wide_queries(rng, n_chrom, span, 30_000, typical=40_000)  # ~90 ids per wide query
    _assert_same_queries(g, o, qc[:6000], qs[:6000], qe[:6000], min_overlaps=(None, 1, 300))
    off_o, ids_o = o.tokenize(qc, qs, qe)
    assert len(ids_o) > 40 * len(qc)
    for rounds, groups in (("1", "1"), ("2", "1"), ("1", "2"), ("2", "2")):
        monkeypatch.setenv("GTARS_TOK_ROUNDS", rounds)
        monkeypatch.setenv("GTARS_TOK_GROUPS", groups)
        off_g, ids_g = g.tokenize(qc, qs, qe)
        assert np.array_equal(off_g, off_o) and np.array_equal(ids_g, ids_o), (rounds, groups)
    monkeypatch.setenv("GTARS_TOK_NARROW", "1")  # the kernels built without the run form (what a launch with a small id buffer takes)
    off_g, ids_g = g.tokenize(qc, qs, qe)
    assert np.array_equal(off_g, off_o) and np.array_equal(ids_g, ids_o)
    monkeypatch.delenv("GTARS_TOK_NARROW")
    monkeypatch.setenv("GTARS_TOK_NO_RUNS", "1")
    off_g, ids_g = g.tokenize(qc, qs, qe)
    assert np.array_equal(off_g, off_o) and np.array_equal(ids_g, ids_o)
    assert g.count_overlaps(qc, qs, qe).tolist() == o.count_overlaps(qc, qs, qe).tolist()


@pytest.mark.parametrize("kind", BOTH)
def test_wide_queries_on_overlapping_universes(ga, monkeypatch, kind):
    """The run form without ascending ends: a ChIP-like universe (overlapping neighbours, 1 % intervals of 5-100 kbp: the search
    key runs ahead of the starts, the run's end is found by stepping), wide queries whose first record has holes in its hit mask,
    and a chromosome with an interval that covers all of it (every tail is stepped through block by block) -- against the oracle
    and against the walked form."""
    from gtars_amd import synth

    u = synth.make_universe(30_000, overlapping=True)
    rng = np.random.default_rng(5)
    order = np.lexsort((u["start"], u["chrom"]))  # (a sorted universe file: ids follow from the position)
    C_, S, E = u["chrom"][order], u["start"][order].copy(), u["end"][order].copy()
    # chromosome 3: one interval over the whole chromosome, in front
    k = np.flatnonzero(C_ == 3)
    S[k[0]] = 0; E[k[0]] = int(E[k].max()) + 10
    g, o = _pair(ga, C_, S, E, n_chrom=synth.N_CHROM, kind=kind)
    nq = 60_000
    qc = rng.integers(0, synth.N_CHROM, nq)
    span = np.array([int(E[C_ == c].max()) if (C_ == c).any() else 1000 for c in range(synth.N_CHROM)])
    qs = (rng.random(nq) * span[qc]).astype(np.int64)
    qe = qs + np.where(rng.random(nq) < 0.5, rng.integers(0, 800, nq), rng.integers(20_000, 3_000_000, nq))
    qc[:300] = 3  # enough queries on the covered chromosome
    _assert_same_queries(g, o, qc[:8000], qs[:8000], qe[:8000], min_overlaps=(None, 1, 200))
    off_o, ids_o = o.tokenize(qc, qs, qe)
    assert len(ids_o) > 4 * nq
    for rounds, groups in (("1", "1"), ("2", "2")):
        monkeypatch.setenv("GTARS_TOK_ROUNDS", rounds)
        monkeypatch.setenv("GTARS_TOK_GROUPS", groups)
        off_g, ids_g = g.tokenize(qc, qs, qe)
        assert np.array_equal(off_g, off_o) and np.array_equal(ids_g, ids_o), (rounds, groups)
    monkeypatch.setenv("GTARS_TOK_NO_RUNS", "1")
    off_g, ids_g = g.tokenize(qc, qs, qe)
    assert np.array_equal(off_g, off_o) and np.array_equal(ids_g, ids_o)


@pytest.mark.parametrize("kind", BOTH)
def test_wide_queries_with_a_short_id_buffer(ga, kind):
    """gtars_tokenize_device with fewer id slots than the batch has hits (hit-heavy batch, wave-wide stores): the offsets and the
    total are complete, the ids up to the capacity are right, nothing is written behind it -- at capacities that cut a wave's
    region, a run and a 256-byte piece in the middle, and at capacity 0 (offsets only)."""
    import torch

    rng = np.random.default_rng(123)
    n_chrom, span = 3, 3_000_000
    C_, S, E = _disjoint_universe(rng, n_chrom, 7_000, span)
    g, o = _pair(ga, C_, S, E, n_chrom=n_chrom, kind=kind)
    qc, qs, qe = _wide_queries(rng, n_chrom, span, 40_000, typical=30_000)
    qc = np.where(qc >= n_chrom, 0, qc)
    off_o, ids_o = o.tokenize(qc, qs, qe)
    total = len(ids_o)
    assert total > 30 * len(qc)
    dev = torch.device("cuda:0")
    d = [torch.from_numpy(np.ascontiguousarray(x, dtype=np.uint32).view(np.int32)).to(dev) for x in (qc, qs, qe)]
    GUARD = 4096
    for cap in (0, 1, 63, 64, 1000, total // 3 + 17, total - 1, total, total + 5):
        off = torch.zeros(len(qc) + 1, dtype=torch.int64, device=dev)
        ids = torch.full((cap + GUARD,), -7, dtype=torch.int32, device=dev)
        args = (d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), len(qc), off.data_ptr(), ids.data_ptr() if cap else 0, cap)
        if 0 < cap < total:  # the call reports the shortfall (status 4, "need <total>") after everything else is in place
            from gtars_amd._lib import CapacityError

            with pytest.raises(CapacityError, match=str(total)):
                g.tokenize_device(*args)
        else:
            assert g.tokenize_device(*args) == total
        assert np.array_equal(off.cpu().numpy().view(np.uint64), off_o)
        got = ids.cpu().numpy()
        k = min(cap, total)
        assert np.array_equal(got[:k].view(np.uint32), ids_o[:k]), cap
        assert (got[k:] == -7).all(), cap


@pytest.mark.parametrize("top_max", ["64", "600"])
def test_run_form_with_padded_units(ga, monkeypatch, top_max):
    """the same with several blocks per search unit (GTARS_TOP_MAX: the chromosomes' block ranges are padded to whole units, a
    query that reaches past a chromosome's last interval lands on a padding block) and a narrow-query majority, so that wide
    queries share their waves with staged ones"""
    monkeypatch.setenv("GTARS_TOP_MAX", top_max)
    rng = np.random.default_rng(int(top_max))
    n_chrom, span = 7, 2_000_000
    C_, S, E = _disjoint_universe(rng, n_chrom, 3_333, span)
    g, o = _pair(ga, C_, S, E, n_chrom=n_chrom)
    qc, qs, qe = _wide_queries(rng, n_chrom, span, 40_000, typical=60_000)
    narrow = rng.random(len(qc)) < 0.97
    qe = np.where(narrow & (qe > qs), np.minimum(qe, qs + 300), qe)
    _assert_same_queries(g, o, qc, qs, qe, min_overlaps=(None,))
    # (few ids per query: the launch takes the kernels built without the run form; once more with them)
    monkeypatch.setenv("GTARS_TOK_WIDE", "1")
    _assert_same_queries(g, o, qc, qs, qe, min_overlaps=(None,))


def test_config3_igd_full_size_properties(ga):
    """BASELINE config 3 at full size (5e7 records, F = 1000, 1e7 queries): size-independent properties of
    the per-file vectors, plus bit-exact parity with the oracle's literal tile walk on six chromosomes (1.2M of the queries)."""
    from gtars_amd import synth

    F = 1000
    db = synth.make_igd_db(50_000_000, F)
    q = synth.make_background_queries(10_000_000)
    g = ga.IgdIndex(db["chrom"], db["start"], db["end"], db["file"], n_chrom=synth.N_CHROM, n_files=F)
    qc, qs, qe = q["chrom"], q["start"], q["end"]
    pair = g.count_set_overlaps(qc, qs, qe, 1)
    binr = g.count_region_hits(qc, qs, qe, 1)
    assert pair.sum() > 0 and (binr <= pair).all() and (binr <= len(qc)).all()
    # additivity over a split of the batch (pairwise and binary are both sums over queries)
    h = len(qc) // 3
    for fn, whole in ((g.count_set_overlaps, pair), (g.count_region_hits, binr)):
        a = fn(qc[:h], qs[:h], qe[:h], 1)
        b = fn(qc[h:], qs[h:], qe[h:], 1)
        assert np.array_equal(a + b, whole)
    # order of the queries does not matter: the (chromosome, start)-sorted batch takes the no-sort path
    order = np.lexsort((qs, qc))
    assert np.array_equal(g.count_set_overlaps(qc[order], qs[order], qe[order], 1), pair)
    assert np.array_equal(g.count_region_hits(qc[order], qs[order], qe[order], 1), binr)
    # a larger min_overlap can only lose hits
    assert (g.count_set_overlaps(qc[:h], qs[:h], qe[:h], 50) <= g.count_set_overlaps(qc[:h], qs[:h], qe[:h], 1)).all()
    # oracle parity on six chromosomes (chr17, chr18, chr19, chr21, chr22, chrY: 12 % of the genome): the literal tile walk over
    # their ~6M records, every query of the batch that lies on them (~1.2M), pairwise and binary
    chroms = np.array([16, 17, 18, 20, 21, 23])
    sel = np.nonzero(np.isin(db["chrom"], chroms))[0]
    o = oracle.Igd()
    o.add_arrays(db["chrom"][sel], db["start"][sel], db["end"][sel], np.zeros(len(sel), dtype=np.int64), db["file"][sel])
    o.n_files = F
    o.finalize()
    qsel = np.nonzero(np.isin(qc, chroms))[0]
    assert len(qsel) >= 1_000_000 and len(np.unique(qc[qsel])) == len(chroms)
    assert np.array_equal(g.count_set_overlaps(qc[qsel], qs[qsel], qe[qsel], 1),
                          o.count_set_overlaps(qc[qsel], qs[qsel], qe[qsel], 1, n_files=F))
    assert np.array_equal(g.count_region_hits(qc[qsel], qs[qsel], qe[qsel], 1),
                          o.count_region_hits(qc[qsel], qs[qsel], qe[qsel], 1, n_files=F))


def test_config4_lola_counts_identities(ga):
    """BASELINE config 4 shape (scaled to 400 DB sets x 25k): support counts through the binary IGD count and
    the contingency kernel satisfy a+b = universe hits, a+c = |user|, a+b+c+d = |universe|, and equal the
    oracle's cells on a sample of the DB sets."""
    from gtars_amd import synth

    F, per = 400, 25_000
    db = synth.make_igd_db(F * per, F, seed=6)
    uni = synth.make_universe(200_000, seed=3)
    rng = np.random.default_rng(9)
    sel = np.sort(rng.choice(len(uni["chrom"]), 20_000, replace=False))
    g = ga.IgdIndex(db["chrom"], db["start"], db["end"], db["file"], n_chrom=synth.N_CHROM, n_files=F)
    uh = g.count_region_hits(uni["chrom"], uni["start"], uni["end"], 1).astype(np.int64)
    sh = g.count_region_hits(uni["chrom"][sel], uni["start"][sel], uni["end"][sel], 1).astype(np.int64)
    import torch

    from gtars_amd._lib import check, lib

    dev = torch.device("cuda:0")
    d_sh, d_uh = torch.from_numpy(sh).to(dev), torch.from_numpy(uh).to(dev)
    cells = [torch.empty(F, dtype=torch.int64, device=dev) for _ in range(4)]
    check(lib.gtars_lola_contingency_device(d_sh.data_ptr(), d_uh.data_ptr(), F, len(sel), len(uni["chrom"]),
                                            *[x.data_ptr() for x in cells], torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    a, b, c, d = (x.cpu().numpy() for x in cells)
    for got, exp in zip((a, b, c, d), oracle.lola_contingency(sh, uh, len(sel), len(uni["chrom"]))):
        assert np.array_equal(got, exp)
    assert np.array_equal(a + b, uh) and (a + c == len(sel)).all() and (a + b + c + d == len(uni["chrom"])).all()
    assert (sh <= uh).all()  # the user set is a subset of the universe
    # oracle parity of the support vectors on the records of 8 DB sets
    keep = np.isin(db["file"], np.arange(8))
    o = oracle.Igd()
    L = oracle.lib()
    for i in np.nonzero(keep)[0]:
        L.orc_igd_add(o._h, int(db["chrom"][i]), int(db["start"][i]), int(db["end"][i]), 0, int(db["file"][i]))
    o.n_files = 8
    o.finalize()
    exp = o.count_region_hits(uni["chrom"][sel], uni["start"][sel], uni["end"][sel], 1, n_files=8)
    assert np.array_equal(sh[:8], exp.astype(np.int64))


def test_config4_full_size_support_vectors(ga):
    """BASELINE config 4 at its full single-GPU size: 2000 DB sets x 25k regions (5e7 records), universe 1e6, user set
    1e5.  The support vectors (binary IGD counts) of the user set and of the universe equal the oracle's on 24 DB sets
    spread over the whole file range (their records only: per-file counts are independent of the other files), the
    pairwise counts too, and the contingency identities hold for all 2000 sets."""
    from gtars_amd import synth

    F, per = 2000, 25_000
    db = synth.make_igd_db(F * per, F, seed=12)
    uni = synth.make_universe(1_000_000, seed=4)
    rng = np.random.default_rng(10)
    sel = np.sort(rng.choice(len(uni["chrom"]), 100_000, replace=False))
    user = {k: uni[k][sel] for k in ("chrom", "start", "end")}
    g = ga.IgdIndex(db["chrom"], db["start"], db["end"], db["file"], n_chrom=synth.N_CHROM, n_files=F)
    uh = g.count_region_hits(uni["chrom"], uni["start"], uni["end"], 1).astype(np.int64)
    sh = g.count_region_hits(user["chrom"], user["start"], user["end"], 1).astype(np.int64)
    pw = g.count_set_overlaps(user["chrom"], user["start"], user["end"], 1).astype(np.int64)
    a, b, c, d = oracle.lola_contingency(sh, uh, len(sel), len(uni["chrom"]))
    assert np.array_equal(a + b, uh) and (a + c == len(sel)).all() and (a + b + c + d == len(uni["chrom"])).all()
    assert (sh <= uh).all() and (sh <= pw).all() and int(sh.sum()) > 0
    sample = np.unique(np.concatenate([[0, 1, F - 1, F - 2], rng.integers(0, F, 20)]))
    remap = np.full(F, -1, dtype=np.int64)
    remap[sample] = np.arange(len(sample))
    keep = np.nonzero(remap[db["file"]] >= 0)[0]
    o = oracle.Igd()
    o.add_arrays(db["chrom"][keep], db["start"][keep], db["end"][keep], np.zeros(len(keep), dtype=int), remap[db["file"][keep]])
    o.n_files = len(sample)
    o.finalize()
    assert np.array_equal(sh[sample], o.count_region_hits(user["chrom"], user["start"], user["end"], 1, n_files=len(sample)).astype(np.int64))
    assert np.array_equal(pw[sample], o.count_set_overlaps(user["chrom"], user["start"], user["end"], 1, n_files=len(sample)).astype(np.int64))
    assert np.array_equal(uh[sample], o.count_region_hits(uni["chrom"], uni["start"], uni["end"], 1, n_files=len(sample)).astype(np.int64))


_HELP_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import gtars_amd, oracle
from gtars_amd import synth
u = synth.make_universe(30_000)
ok = True
for nq in (900, 120_000, 1_300_000):
    q = synth.make_queries(u, nq, seed=nq)
    g = gtars_amd.OverlapIndex(u["chrom"], u["start"], u["end"], n_chrom=synth.N_CHROM)
    o = oracle.Index(u["chrom"], u["start"], u["end"], None, n_chrom=synth.N_CHROM)
    off_g, ids_g = g.tokenize(q["chrom"], q["start"], q["end"])
    off_o, ids_o = o.tokenize(q["chrom"], q["start"], q["end"])
    ok = ok and np.array_equal(off_g, off_o) and np.array_equal(ids_g, ids_o)
print("HELP_PARITY", ok)
"""


VARIANTS = {  # launch geometries of the fused tokenizer (tokenize_lds.hip): forced through the environment
    "tile4096": {"GTARS_TOK_ROUNDS": "1", "GTARS_TOK_GROUPS": "1"},
    "tile8192": {"GTARS_TOK_ROUNDS": "2", "GTARS_TOK_GROUPS": "1"},
    "two_groups_tile2048": {"GTARS_TOK_ROUNDS": "1", "GTARS_TOK_GROUPS": "2"},
    "two_groups_tile4096": {"GTARS_TOK_ROUNDS": "2", "GTARS_TOK_GROUPS": "2"},
}


@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_lookback_helps_itself_instead_of_waiting(ga, variant):
    """The tokenizer never depends on a predecessor tile being resident: a look-back that has polled an unpublished
    predecessor `spin_limit` times counts that tile itself.  With the limit forced to 0 EVERY wait takes that path
    (own subprocess: the limit is read once per process); results stay bit-exact."""
    import os, subprocess, sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GTARS_TOK_SPIN_LIMIT="0", **VARIANTS[variant])
    r = subprocess.run([sys.executable, "-c", _HELP_SCRIPT % root], env=env, capture_output=True, text=True, timeout=600)
    assert "HELP_PARITY True" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_tokenizer_launch_geometries_agree_with_the_oracle(ga, monkeypatch, variant):
    """Every launch geometry (one or two rounds per lane, one or two wave groups per workgroup) on: a sorted universe (ids from
    the position), shuffled ids (explicit id records), a dense universe (tiles whose hits overflow the LDS stage take
    the direct path), AIList order, a minimum-overlap filter, batch sizes around the tile sizes, and a caller
    capacity smaller than the result."""
    import torch
    from gtars_amd import synth

    for k, v in VARIANTS[variant].items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(77)
    u = synth.make_universe(20_000)
    shuffled = rng.permutation(len(u["chrom"])).astype(np.uint32)
    cases = [(u["chrom"], u["start"], u["end"], None, KIND_BITS), (u["chrom"], u["start"], u["end"], shuffled, KIND_BITS),
             (u["chrom"], u["start"], u["end"], shuffled, KIND_AILIST)]
    dense_s = np.sort(rng.integers(0, 200_000, 30_000)).astype(np.uint32)
    dense = (np.zeros(30_000, dtype=np.uint32), dense_s, dense_s + rng.integers(50, 400, 30_000).astype(np.uint32), None, KIND_BITS)
    for c, s, e, val, kind in cases + [dense]:
        n_chrom = int(c.max()) + 1
        g, o = _pair(ga, c, s, e, val, n_chrom=n_chrom, kind=kind)
        for nq in (1, 3839, 3841, 8193, 50_001):
            if c is dense[0]:
                qc = np.zeros(nq, dtype=np.uint32)
                qs = rng.integers(0, 200_000, nq).astype(np.uint32)
                qe = qs + rng.integers(1, 300, nq).astype(np.uint32)
            else:
                q = synth.make_queries(u, nq, seed=nq)
                qc, qs, qe = q["chrom"].copy(), q["start"], q["end"]
                qc[::53] = UNK
            off_o, ids_o = o.tokenize(qc, qs, qe)
            off_g, ids_g = g.tokenize(qc, qs, qe)
            assert np.array_equal(off_g, off_o) and np.array_equal(ids_g, ids_o), (variant, kind, nq)
            fo = o.find_overlaps_regions(qc, qs, qe, 20)
            fg = g.find_overlaps(qc, qs, qe, 20)
            assert all(np.array_equal(a, b) for a, b in zip(fg, fo)), (variant, kind, nq, "min_overlap")
        # capacity below the result: offsets complete, ids written up to the capacity, nothing beyond it
        dev = torch.device("cuda:0")
        d = [torch.from_numpy(x.view(np.int32)).to(dev) for x in (qc, qs, qe)]
        h_all = len(ids_o)
        capn = h_all // 2
        offsets = torch.empty(nq + 1, dtype=torch.int64, device=dev)
        ids = torch.full((h_all + 8,), -7, dtype=torch.int32, device=dev)
        with pytest.raises(ga.CapacityError):
            g.tokenize_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, offsets.data_ptr(), ids.data_ptr(), capn,
                              torch.cuda.current_stream().cuda_stream, sync=True)
        assert np.array_equal(offsets.cpu().numpy().astype(np.uint64), off_o.astype(np.uint64))
        assert np.array_equal(ids[:capn].cpu().numpy().view(np.uint32), ids_o[:capn])
        assert bool((ids[capn:] == -7).all())


def test_scan_epoch_wraps_after_16k_launches(ga):
    """The chained-scan workspace is never cleared between launches: granules carry a 14-bit launch epoch.
    After 16383 launches on one stream the epoch wraps (one memset); results before, at and after the wrap
    stay bit-exact, also when batch sizes (and thus tile counts and ticket ranges) change in between."""
    import torch
    from gtars_amd import synth

    u = synth.make_universe(5_000)
    g, o = _pair(ga, u["chrom"], u["start"], u["end"], n_chrom=synth.N_CHROM)
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    batches = []
    for nq, seed in ((9_000, 1), (300_001, 2)):   # one tile per workgroup / more tiles than workgroups
        q = synth.make_queries(u, nq, seed=seed)
        qc, qs, qe = (torch.from_numpy(q[k].view(np.int32)).to(dev) for k in ("chrom", "start", "end"))
        off_o, ids_o = o.tokenize(q["chrom"], q["start"], q["end"])
        offsets = torch.empty(len(q["chrom"]) + 1, dtype=torch.int64, device=dev)
        ids = torch.empty(len(ids_o) + 8, dtype=torch.int32, device=dev)
        batches.append((qc, qs, qe, offsets, ids, off_o, ids_o))

    def run(b, sync):
        qc, qs, qe, offsets, ids, off_o, ids_o = b
        h = g.tokenize_device(qc.data_ptr(), qs.data_ptr(), qe.data_ptr(), qc.numel(), offsets.data_ptr(), ids.data_ptr(),
                              ids.numel(), st, sync=sync)
        if sync:
            assert h == len(ids_o)
            assert np.array_equal(offsets.cpu().numpy().view(np.uint64), off_o)
            assert np.array_equal(ids[:h].cpu().numpy().view(np.uint32), ids_o)

    for i in range(16_600):
        big = (i % 997 == 0)
        check = i in (0, 1, 16_380, 16_381, 16_382, 16_383, 16_384, 16_385, 16_599) or big
        run(batches[1] if big else batches[0], check)
    run(batches[1], True)


def test_key_space_wider_than_32_bits(ga):
    """Chromosome coordinates are u32, but the search key space concatenates the chromosomes: with three
    chromosomes whose ends reach 4e9 it is ~1.2e10 wide.  The LDS kernel handles that (64-bit keys)."""
    rng = np.random.default_rng(64)
    n_chrom, n = 3, 60_000
    c = rng.integers(0, n_chrom, n)
    s = rng.integers(0, 4_000_000_000, n, dtype=np.uint64)
    e = np.minimum(s + rng.integers(1, 100_000, n).astype(np.uint64), 0xFFFFFFFF)
    g, o = _pair(ga, c, s, e, n_chrom=n_chrom)
    nq = 100_000
    qc = rng.integers(0, n_chrom + 1, nq)
    qc = np.where(qc >= n_chrom, UNK, qc)
    qs = rng.integers(0, 4_100_000_000, nq, dtype=np.uint64)
    qe = np.minimum(qs + rng.integers(0, 300_000, nq).astype(np.uint64), 0xFFFFFFFF)
    _lib = ga._lib
    _lib.lib.gtars_prof_reset()
    _lib.lib.gtars_prof_enable(1)
    off_g, ids_g = g.tokenize(qc, qs, qe)
    names = set(_lib.prof_read())
    _lib.lib.gtars_prof_enable(0)
    off_o, ids_o = o.tokenize(qc, qs, qe)
    assert np.array_equal(off_g, off_o) and np.array_equal(ids_g, ids_o)
    assert names & {"k_tok_lds", "k_tok_wave"}, names
    assert np.array_equal(g.count_overlaps(qc, qs, qe), o.count_overlaps(qc, qs, qe))


def test_dense_database_hundreds_of_hits_per_query(ga):
    """A ChIP-like dense database: 500+ hits per query.  The paths that used to be quadratic in the hits of a query --
    per-segment sort + unique (IndexedRegionSet::find_overlaps), the per-query binary IGD count, the "first occurrence
    of a value" walk of find_overlaps_regionset / count_overlaps_per_query -- stay exact: heap sort beyond 24 hits,
    pme_file for min_overlap == 1 (the scanned-prefix look-up for min_overlap > 1), values known to be distinct
    (from_single_region_set) or not (a database whose values repeat)."""
    rng = np.random.default_rng(2024)
    n, span = 40_000, 60_000
    c = np.zeros(n, dtype=np.int64)
    s = rng.integers(0, span, n)
    e = s + rng.integers(1, 2_000, n)
    qn = 300
    qc = np.zeros(qn, dtype=np.int64)
    qs = rng.integers(0, span, qn)
    qe = qs + rng.integers(100, 1_500, qn)
    # IndexedRegionSet: sorted unique source indices per query
    for kind in BOTH:
        g, o = _pair(ga, c, s, e, n_chrom=1, kind=kind)
        og, ig = g.find_overlap_indices(qc, qs, qe)
        oo, io = o.irs_find_overlaps(c, s, e, qc, qs, qe)
        assert og.tolist() == oo.tolist() and ig.tolist() == io.tolist()
        assert np.diff(og.astype(np.int64)).max() >= 500
    # IGD, distinct values (value = source index): pairs in the reference's walk order, per-query counts
    f = rng.integers(0, 7, n)
    g, o = _igd_pair(ga, c, s, e, f, np.arange(n), n_chrom=1, n_files=7)
    for mo in (1, 40):
        gq, gs = g.find_overlaps_regionset(qc, qs, qe, mo)
        oq, os_ = o.find_overlaps_regionset(qc, qs, qe, mo)
        assert gq.tolist() == oq.tolist() and gs.tolist() == os_.tolist()
        assert g.count_overlaps_per_query(qc, qs, qe, mo).tolist() == o.count_overlaps_per_query(qc, qs, qe, mo).tolist()
        assert g.count_region_hits(qc, qs, qe, mo).tolist() == o.count_region_hits(qc, qs, qe, mo, n_files=7).tolist()
        assert g.count_set_overlaps(qc, qs, qe, mo).tolist() == o.count_set_overlaps(qc, qs, qe, mo, n_files=7).tolist()
    # values that repeat: the de-duplicating walk
    g2, o2 = _igd_pair(ga, c, s, e, f, rng.integers(0, 5_000, n), n_chrom=1, n_files=7)
    gq, gs = g2.find_overlaps_regionset(qc[:60], qs[:60], qe[:60], 1)
    oq, os_ = o2.find_overlaps_regionset(qc[:60], qs[:60], qe[:60], 1)
    assert gq.tolist() == oq.tolist() and gs.tolist() == os_.tolist()
    assert g2.count_overlaps_per_query(qc[:60], qs[:60], qe[:60], 1).tolist() == o2.count_overlaps_per_query(qc[:60], qs[:60], qe[:60], 1).tolist()


@pytest.mark.parametrize("hog_wgs,hog_lds", [(200, 100 * 1024), (512, 60 * 1024)])
def test_tokenize_async_while_another_stream_holds_cus(ga, hog_wgs, hog_lds):
    """An ASYNCHRONOUS gtars_tokenize_device(total_hits = NULL) launch that shares the GPU with foreign work: a kernel on a
    second stream occupies most CUs (one 1024-thread workgroup with 100 KB of LDS per CU, or two with 60 KB) for several
    milliseconds while tokenizer launches of 1..40 tiles per workgroup are queued on another stream.  Whatever part of the
    tokenizer's grid is resident at a time, every launch must complete with the exact offsets and ids (tiles are handed
    out by ticket, and a look-back that cannot see a predecessor counts that tile itself)."""
    import torch
    from gtars_amd import synth
    from gtars_amd._lib import check, lib

    u = synth.make_universe(50_000)
    g, o = _pair(ga, u["chrom"], u["start"], u["end"], n_chrom=synth.N_CHROM)
    dev = torch.device("cuda:0")
    s_tok, s_hog = torch.cuda.Stream(), torch.cuda.Stream()
    cases = []
    for nq in (300_000, 1_000_000, 9_000_000):
        q = synth.make_queries(u, min(nq, 1_000_000), seed=nq)
        rep = max(nq // len(q["chrom"]), 1)
        qc, qs, qe = (torch.from_numpy(np.tile(q[k], rep).view(np.int32)).to(dev) for k in ("chrom", "start", "end"))
        off_o, ids_o = o.tokenize(q["chrom"], q["start"], q["end"])
        n = qc.numel()
        offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        ids = torch.zeros(len(ids_o) * rep + 8, dtype=torch.int32, device=dev)
        cases.append((qc, qs, qe, n, offsets, ids, off_o, ids_o, rep))
    torch.cuda.synchronize()
    for round_ in range(3):
        check(lib.gtars_debug_occupy_device(s_hog.cuda_stream, hog_wgs, hog_lds, 4000))
        for qc, qs, qe, n, offsets, ids, off_o, ids_o, rep in cases:
            offsets.zero_()
            ids.zero_()
            s_tok.wait_stream(torch.cuda.current_stream())  # (the clears run on the current stream: they must not overtake the launch)
            with torch.cuda.stream(s_tok):
                g.tokenize_device(qc.data_ptr(), qs.data_ptr(), qe.data_ptr(), n, offsets.data_ptr(), ids.data_ptr(), ids.numel(),
                                  s_tok.cuda_stream, sync=False)
        torch.cuda.synchronize()
        for qc, qs, qe, n, offsets, ids, off_o, ids_o, rep in cases:
            got_off = offsets.cpu().numpy().view(np.uint64)
            h1 = len(ids_o)
            assert int(got_off[-1]) == h1 * rep
            nq1 = len(off_o) - 1
            assert np.array_equal(got_off[: nq1 + 1], off_o)
            assert np.array_equal(ids[:h1].cpu().numpy().view(np.uint32), ids_o)
            if rep > 1:  # every repetition of the base batch: same counts, same ids
                assert np.array_equal(np.diff(got_off.astype(np.int64)).reshape(rep, nq1), np.tile(np.diff(off_o.astype(np.int64)), (rep, 1)))
                assert np.array_equal(ids[h1 * (rep - 1): h1 * rep].cpu().numpy().view(np.uint32), ids_o)


# ------------------------------------------------------------ Bits::insert / Bits::seek behind the C ABI


def test_bits_insert_and_seek_match_the_reference_rules(ga):
    """gtars_index_insert / gtars_index_seek vs the literal list restatement (oracle.MutableBits): the doc examples of
    bits.rs:193-206 and 351-361, then random inserts (equal keys, wider-than-max_len intervals, two chromosomes) with
    sorted and unsorted seek sequences carrying one cursor, and the device path (tokenize) on the index after inserts."""
    g = ga.OverlapIndex(np.zeros(2, dtype=np.uint32), np.array([0, 6], dtype=np.uint32), np.array([5, 10], dtype=np.uint32),
                        np.array([1, 2], dtype=np.uint32), n_chrom=1)
    g.insert(0, 0, 20, 5)
    assert len(g) == 3 and g.max_len(0) == 20
    off, ids = g.tokenize(np.zeros(1, dtype=np.uint32), np.array([1], dtype=np.uint32), np.array([3], dtype=np.uint32))
    assert ids.tolist() == [1, 5]  # {0,5,1} then {0,20,5}
    xs = np.arange(0, 100, 5, dtype=np.uint32)
    g = ga.OverlapIndex(np.zeros(len(xs), dtype=np.uint32), xs, xs + 2, n_chrom=1)
    cur = 0
    for x in xs.tolist():
        vals, cur = g.seek(0, x, x + 2, cur)
        assert len(vals) == 1
    rng = np.random.default_rng(9)
    n = 400
    c = rng.integers(0, 2, n).astype(np.uint32)
    s = rng.integers(0, 3000, n).astype(np.uint32)
    e = (s + rng.integers(0, 100, n)).astype(np.uint32)
    v = np.arange(n, dtype=np.uint32)
    g = ga.OverlapIndex(c, s, e, v, n_chrom=2)
    refs = [oracle.MutableBits([(int(s[i]), int(e[i]), int(v[i])) for i in range(n) if c[i] == ch]) for ch in (0, 1)]
    for k in range(60):
        ch = int(rng.integers(0, 2))
        if k % 3 == 0:  # an equal (start, end) key: goes in front of the existing ones
            a, b = refs[ch].intervals[int(rng.integers(0, len(refs[ch])))][:2]
        else:
            a = int(rng.integers(0, 3000))
            b = a + int(rng.integers(0, 400))
        g.insert(ch, a, b, 10_000 + k)
        refs[ch].insert(a, b, 10_000 + k)
    for ch in (0, 1):
        st, en, va = g.stored(ch)
        assert list(zip(st.tolist(), en.tolist(), va.tolist())) == refs[ch].intervals
        assert g.max_len(ch) == refs[ch].max_len
        for sorted_queries in (True, False):
            qs = rng.integers(0, 3200, 300)
            if sorted_queries:
                qs = np.sort(qs)
            cur_g = cur_o = 0
            for q in qs.tolist():
                w = int(rng.integers(0, 200))
                vals, cur_g = g.seek(ch, q, q + w, cur_g)
                hits, cur_o = refs[ch].seek(q, q + w, cur_o)
                assert cur_g == cur_o and vals.tolist() == [t[2] for t in hits]
    # the device structures were rebuilt: the kernel path sees the inserted intervals
    qs = rng.integers(0, 3200, 500).astype(np.uint32)
    qe = (qs + rng.integers(0, 200, 500)).astype(np.uint32)
    qc = rng.integers(0, 2, 500).astype(np.uint32)
    off, ids = g.tokenize(qc, qs, qe)
    for i in range(500):
        assert ids[off[i]:off[i + 1]].tolist() == [t[2] for t in refs[int(qc[i])].find(int(qs[i]), int(qe[i]))]
    with pytest.raises(ValueError):  # GTARS_ERR_INVALID_ARG: insert is a Bits operation
        ga.OverlapIndex(c, s, e, v, n_chrom=2, kind=KIND_AILIST).insert(0, 1, 2, 3)


def test_igd_two_level_partition_with_a_sparse_tail(ga, monkeypatch):
    """The two-level, LDS-reordered partition (k_split_pass) on a skewed batch: 1.2M queries, most of them inside one
    100-kb window, 10k spread over the whole genome.  In the second pass a tile of the sparse tail spans far more than
    the 1024-key window, so its elements take the direct-slot path.  6000 queries have NO owner tile (unknown chromosome,
    start past every record of the chromosome, start >= end): the partition drops them in its second pass, and the first
    pass has to carry them (a first pass that drops them leaves the tail of its output to whatever an earlier call wrote
    there: round 3's soak found exactly that).  Three independent routes must agree: the shuffled batch (partitioned), the
    same batch in (chromosome, start) order (no partition at all) and the per-query kernel (no sweep), for pairwise and
    binary counts; the partitioned call runs twice with a different batch in between, so stale workspace would show."""
    from gtars_amd import synth

    F = 50
    db = synth.make_igd_db(45_000_000, F, seed=21)
    g = ga.IgdIndex(db["chrom"], db["start"], db["end"], db["file"], n_chrom=synth.N_CHROM, n_files=F)
    del db
    rng = np.random.default_rng(8)
    n_hot, n_tail = 1_190_000, 10_000
    bg = synth.make_background_queries(n_tail, seed=5)
    n_lost = 2000
    lost_c = np.concatenate([np.full(n_lost, 0xFFFFFFFF, dtype=np.uint32), np.zeros(2 * n_lost, dtype=np.uint32)])
    lost_s = np.concatenate([rng.integers(0, 10**8, n_lost), rng.integers(3 * 10**8, 4 * 10**8, n_lost), rng.integers(10**6, 10**8, n_lost)])
    lost_e = np.concatenate([lost_s[:2 * n_lost] + 300, lost_s[2 * n_lost:] - rng.integers(0, 50, n_lost)])
    qc = np.concatenate([np.zeros(n_hot, dtype=np.uint32), bg["chrom"], lost_c])
    hs = rng.integers(5_000_000, 5_100_000, n_hot)
    qs = np.concatenate([hs, bg["start"], lost_s]).astype(np.uint32)
    qe = np.concatenate([hs + rng.integers(1, 500, n_hot), bg["end"], lost_e]).astype(np.uint32)
    sh = rng.permutation(len(qc))
    qc, qs, qe = qc[sh], qs[sh], qe[sh]
    order = np.lexsort((qs, np.where(qs >= qe, 0xFFFFFFFF, qc)))  # rejected queries last: the batch stays in owner order
    other = synth.make_background_queries(1_100_000, seed=6)
    for count in (g.count_set_overlaps, g.count_region_hits):
        shuffled = count(qc, qs, qe, 1)
        count(other["chrom"], other["start"], other["end"], 1)  # leaves its own intermediate data in the workspace
        assert np.array_equal(count(qc, qs, qe, 1), shuffled)
        in_order = count(qc[order], qs[order], qe[order], 1)
        monkeypatch.setenv("GTARS_IGD_SWEEP_MIN", "100000000")
        per_query = count(qc, qs, qe, 1)
        monkeypatch.delenv("GTARS_IGD_SWEEP_MIN")
        assert int(shuffled.sum()) > n_hot  # the hot window is covered
        assert np.array_equal(shuffled, in_order) and np.array_equal(shuffled, per_query)
        # the hot tile owns ~1.19M queries: it is served in ~290 parts by as many work items of the sweep (HeavyBins); the same
        # vectors with one workgroup per tile
        monkeypatch.setenv("GTARS_IGD_NO_HEAVY_PARTS", "1")
        assert np.array_equal(count(qc, qs, qe, 1), shuffled)
        assert np.array_equal(count(qc[order], qs[order], qe[order], 1), shuffled)
        monkeypatch.delenv("GTARS_IGD_NO_HEAVY_PARTS")


def _random_query_set(rng, n, n_chrom, span, wmax, spoiled=0.02):
    """n queries, a few of them unknown chromosomes, inverted, or negative as i32 (rejected or clamped by igd.rs:514-517)"""
    qc = rng.integers(0, n_chrom, n).astype(np.uint32)
    qs = rng.integers(0, span, n).astype(np.int64)
    qe = qs + rng.integers(1, wmax, n)
    bad = rng.random(n) < spoiled
    kind = rng.integers(0, 3, n)
    qc = np.where(bad & (kind == 0), UNK, qc).astype(np.uint32)
    qe = np.where(bad & (kind == 1), qs - 5, qe)
    neg = bad & (kind == 2)
    qe = np.where(neg, rng.integers(1, wmax, n), qe)  # (short: the clamped query is [0, end))
    qs = np.where(neg, 2**32 - 1 - rng.integers(0, 1000, n), qs)  # negative as i32: clamped to 0
    return qc, (qs % 2**32).astype(np.uint32), (qe % 2**32).astype(np.uint32)


@pytest.mark.parametrize("n_db,sizes", [(150_000, (400_000, 60_000)), (2_300_000, (1_100_000, 90_000, 0, 7_000)),
                                        (150_000, (300_000, 20_000, 20_000, 20_000, 250_000, 1_000)), (150_000, (3_000, 500))])
def test_igd_query_sets_share_one_pass(ga, n_db, sizes):
    """gtars_igd_count_sets: the count step of run_lola (enrichment.rs:198-221 -- count_region_hits of the universe and of every
    user set over the same Igd).  Up to four sets share one sweep of the database (the partition tags the pairs, one row of
    counters per set); every row must equal what the oracle returns for that set ALONE, pairwise and binary, min_overlap 1
    (pme_file form) and 3 (credited-file-list form); how many passes were shared is read from the profiling facts, not from
    a clock.  Cases: one- and two-level partitions, an empty set in the middle, more than four sets (two groups), sets
    below the sweep's crossover (counted set by set)."""
    rng = np.random.default_rng(n_db + len(sizes))
    n_chrom, F, span = 3, 40, 30_000_000 if n_db < 1_000_000 else 240_000_000  # ~15-30 overlapping records per query
    c = rng.integers(0, n_chrom, n_db)
    s = rng.integers(0, span, n_db)
    e = s + rng.integers(1, 3_000, n_db)
    f = rng.integers(0, F, n_db)
    g, o = _igd_pair(ga, c, s, e, f, np.arange(n_db), n_chrom=n_chrom, n_files=F)
    sets = [_random_query_set(rng, n, n_chrom, span + 5_000, 800) for n in sizes]
    _lib = ga._lib
    for binary, mo in ((True, 1), (False, 1), (False, 3), (True, 3)):
        ref = o.count_region_hits if binary else o.count_set_overlaps
        want = np.stack([ref(qc, qs, qe, mo, n_files=F) for qc, qs, qe in sets])
        g.count_sets(sets[:1], mo, binary)  # builds pme_file outside the profiled call
        _lib.lib.gtars_prof_reset()
        _lib.lib.gtars_prof_enable(1)
        got = g.count_sets(sets, mo, binary)
        prof = _lib.prof_read()
        _lib.lib.gtars_prof_enable(0)
        assert np.array_equal(got, want), (binary, mo, np.argwhere(got != want)[:5])
        shared = prof.get("igd_sets_shared_pass", {"launches": 0})["launches"]
        if sizes == (3_000, 500):
            assert shared == 0  # below the crossover: the per-query kernel, set by set
        elif len(sizes) == 6:
            assert shared == 2  # sets 0-3 and sets 4-5
        else:
            assert shared == 1
    # the same rows again set by set (what the shared pass replaces)
    os.environ["GTARS_IGD_SWEEP_MIN"] = "1"
    ga.reload_env()
    try:
        for k, (qc, qs, qe) in enumerate(sets):
            assert np.array_equal(g.count_region_hits(qc, qs, qe, 1), o.count_region_hits(qc, qs, qe, 1, n_files=F)), k
    finally:
        del os.environ["GTARS_IGD_SWEEP_MIN"]
        ga.reload_env()


def test_igd_query_sets_in_order_are_swept_as_they_arrive(ga, monkeypatch):
    """Round 5: several query sets that are each in (chromosome, start) order -- what a LOLA call's universe and user sets are
    (enrichment.rs:198-215 counts them over the same Igd) -- are not partitioned: the sweep serves a tile's queries set after
    set from one row of tile ranges per set.  Every row must equal the oracle's count for that set alone; which continuation the
    device took is read from the profiling facts.  Cases: two to four sets, an empty set in the middle, rejected queries at a
    set's end (they sort last), one set out of order (-> the partition, same rows), the credited-file-list form of binary counts
    (min_overlap 3: always partitioned), a set with far more queries in one tile than the average (no heavy-tile parts in the
    in-order form: the tile serves them all), and the forced partition."""
    monkeypatch.setenv("GTARS_IGD_SWEEP_MIN", "1")
    ga.reload_env()
    try:
        rng = np.random.default_rng(77)
        n_db, n_chrom, F, span = 300_000, 3, 40, 30_000_000
        c = rng.integers(0, n_chrom, n_db)
        s = rng.integers(0, span, n_db)
        e = s + rng.integers(1, 3_000, n_db)
        f = rng.integers(0, F, n_db)
        g, o = _igd_pair(ga, c, s, e, f, np.arange(n_db), n_chrom=n_chrom, n_files=F)

        def ordered(n, dense=False):
            qc, qs, qe = _random_query_set(rng, n, n_chrom, span + 5_000, 800, spoiled=0.0)
            if dense:  # a third of the set inside one 20-kb window
                k = n // 3
                qc[:k] = 1
                qs[:k] = 12_000_000 + rng.integers(0, 20_000, k)
                qe[:k] = qs[:k] + rng.integers(1, 800, k)
            order = np.lexsort((qs, qc))
            qc, qs, qe = qc[order], qs[order], qe[order]
            tail = 7  # rejected queries behind the last real one: unknown chromosome / inverted
            return (np.concatenate([qc, np.full(tail, UNK, dtype=np.uint32)]), np.concatenate([qs, np.arange(tail, dtype=np.uint32) + 5]),
                    np.concatenate([qe, np.arange(tail, dtype=np.uint32) + 9]))

        empty = (np.zeros(0, dtype=np.uint32),) * 3
        cases = {"two": [ordered(400_000), ordered(50_000)], "four with an empty one": [ordered(120_000), empty, ordered(60_000), ordered(90_000)],
                 "dense tile": [ordered(300_000, dense=True), ordered(40_000)]}
        _lib = ga._lib

        def took(prof):
            return {k: v["launches"] for k, v in prof.items() if k.startswith("igd_batch_")}

        def run(sets, mo, binary):
            _lib.lib.gtars_prof_reset()
            _lib.lib.gtars_prof_enable(1)
            got = g.count_sets(sets, mo, binary)
            prof = _lib.prof_read()
            _lib.lib.gtars_prof_enable(0)
            return got, took(prof), prof.get("igd_sets_shared_pass", {"launches": 0})["launches"]

        g.count_sets(cases["two"][:1], 1, True)  # builds pme_file outside the profiled calls
        for name, sets in cases.items():
            for binary, mo in ((True, 1), (False, 1), (False, 3), (True, 3)):
                ref = o.count_region_hits if binary else o.count_set_overlaps
                want = np.stack([ref(qc, qs, qe, mo, n_files=F) for qc, qs, qe in sets])
                got, path, shared = run(sets, mo, binary)
                assert np.array_equal(got, want), (name, binary, mo, np.argwhere(got != want)[:5])
                assert shared == 1, (name, binary, mo)
                # (binary counts with min_overlap 3 keep a list of credited files per query: that form is always partitioned)
                assert path == ({"igd_batch_partitioned": 1} if (binary and mo == 3) else {"igd_batch_in_owner_order": 1}), (name, binary, mo, path)
        # one set out of order: the partition, same rows
        a, b = cases["two"]
        sh = rng.permutation(len(b[0]))
        mixed = [a, tuple(x[sh] for x in b)]
        want = np.stack([o.count_region_hits(qc, qs, qe, 1, n_files=F) for qc, qs, qe in mixed])
        got, path, _ = run(mixed, 1, True)
        assert np.array_equal(got, want) and path == {"igd_batch_partitioned": 1}, path
        # ... and the switch that always partitions
        monkeypatch.setenv("GTARS_IGD_SETS_ALWAYS_PARTITION", "1")
        ga.reload_env()
        want = np.stack([o.count_region_hits(qc, qs, qe, 1, n_files=F) for qc, qs, qe in cases["two"]])
        got, path, _ = run(cases["two"], 1, True)
        assert np.array_equal(got, want) and path == {"igd_batch_partitioned": 1}, path
    finally:
        monkeypatch.delenv("GTARS_IGD_SETS_ALWAYS_PARTITION", raising=False)
        monkeypatch.delenv("GTARS_IGD_SWEEP_MIN", raising=False)
        ga.reload_env()


def test_igd_query_sets_argument_checks(ga):
    rng = np.random.default_rng(5)
    g = ga.IgdIndex(np.zeros(10, dtype=np.uint32), np.arange(10) * 10, np.arange(10) * 10 + 5, np.zeros(10, dtype=np.uint32), n_chrom=1, n_files=1)
    assert g.count_sets([], 1, True).shape == (0, 1)
    assert g.count_sets([(np.zeros(0), np.zeros(0), np.zeros(0))] * 3, 1, True).tolist() == [[0], [0], [0]]
    q = (np.zeros(4, dtype=np.uint32), np.array([0, 10, 20, 200], dtype=np.uint32), np.array([6, 11, 22, 300], dtype=np.uint32))
    assert g.count_sets([q, q], 1, False).tolist() == [[3], [3]]
    assert g.count_sets([q, q], 0, False).tolist() == [g.count_set_overlaps(*q, 0).tolist()] * 2  # min_overlap < 1: set by set
    import ctypes as C
    off = np.array([1, 4], dtype=np.uint64)
    hits = np.zeros(1, dtype=np.uint64)
    lib, ptr = ga._lib.lib, ga._lib.ptr
    assert lib.gtars_igd_count_sets(g._h, ptr(q[0]), ptr(q[1]), ptr(q[2]), ptr(off), 1, 1, 0, ptr(hits)) != 0  # set_off[0] != 0
    off = np.array([0, 4, 2], dtype=np.uint64)
    assert lib.gtars_igd_count_sets(g._h, ptr(q[0]), ptr(q[1]), ptr(q[2]), ptr(off), 2, 1, 0, ptr(hits)) != 0  # decreasing


def test_igd_per_query_counts_with_one_giant_record(ga, monkeypatch):
    """The per-query kernel (batches below the sweep's crossover) starts a query's scan at the first record whose prefix-max end
    is > q_start (IgdTiles::pm) instead of at lower_bound(q_start - the chromosome's longest record): with one 60-Mbp record
    in the database the old start made every query of that chromosome walk ~half the chromosome.  Same vectors as the oracle
    and as the old start (GTARS_IGD_NO_PM_START), pairwise and binary, min_overlap 1 and 4."""
    rng = np.random.default_rng(99)
    n, F, span = 300_000, 25, 100_000_000
    c = rng.integers(0, 2, n)
    s = rng.integers(0, span, n)
    e = s + rng.integers(1, 900, n)
    f = rng.integers(0, F, n)
    c[7], s[7], e[7] = 0, 20_000_000, 80_000_000  # the giant
    g, o = _igd_pair(ga, c, s, e, f, np.arange(n), n_chrom=2, n_files=F)
    qc, qs, qe = _random_query_set(rng, 20_000, 2, span + 2_000, 700)
    for mo in (1, 4):
        want_p = o.count_set_overlaps(qc, qs, qe, mo, n_files=F)
        want_b = o.count_region_hits(qc, qs, qe, mo, n_files=F)
        assert np.array_equal(g.count_set_overlaps(qc, qs, qe, mo), want_p)
        assert np.array_equal(g.count_region_hits(qc, qs, qe, mo), want_b)
        monkeypatch.setenv("GTARS_IGD_NO_PM_START", "1")
        assert np.array_equal(g.count_set_overlaps(qc, qs, qe, mo), want_p)
        assert np.array_equal(g.count_region_hits(qc, qs, qe, mo), want_b)
        monkeypatch.delenv("GTARS_IGD_NO_PM_START")
    monkeypatch.setenv("GTARS_IGD_NO_PME", "1")  # the credited-file form of the binary count
    assert np.array_equal(g.count_region_hits(qc, qs, qe, 1), o.count_region_hits(qc, qs, qe, 1, n_files=F))


@pytest.mark.parametrize("piece_bp", [None, 64])
def test_igd_counts_on_databases_with_long_records(ga, monkeypatch, piece_bp):
    """Databases with records much longer than most (broad peaks, one 30-Mbp record) are counted through a second index of
    PIECES for min_overlap == 1 (api.hip build_pieces_view: long records cut at multiples of the piece length, a continuation piece
    counts only if it holds the query's start -- the reference's own tile rule, igd.rs:109-153 / 812-817).  Every form must
    equal the oracle's literal tile walk over the ORIGINAL records: sweep and per-query kernels, pairwise and binary, several
    query sets in one pass, and min_overlap 3 (served by the flat layout).  piece_bp = 64: the same with tiny pieces, so that
    almost every record is cut."""
    if piece_bp:
        monkeypatch.setenv("GTARS_IGD_PIECE_BP", str(piece_bp))
    rng = np.random.default_rng(2024)
    n, F, span = 120_000, 30, 60_000_000
    c = rng.integers(0, 2, n)
    s = rng.integers(0, span, n)
    w = rng.integers(1, 900, n)
    wide = rng.random(n) < 0.03
    w = np.where(wide, rng.integers(5_000, 300_000, n), w)
    e = s + w
    c[11], s[11], e[11] = 1, 10_000_000, 40_000_000
    f = rng.integers(0, F, n)
    g, o = _igd_pair(ga, c, s, e, f, np.arange(n), n_chrom=2, n_files=F)
    big = _random_query_set(rng, 150_000, 2, span + 2_000, 700)       # above the sweep's crossover
    small = _random_query_set(rng, 9_000, 2, span + 2_000, 5_000)     # per-query kernel
    _lib = ga._lib
    for qc, qs, qe in (big, small):
        _lib.lib.gtars_prof_reset()
        _lib.lib.gtars_prof_enable(1)
        got_p, got_b = g.count_set_overlaps(qc, qs, qe, 1), g.count_region_hits(qc, qs, qe, 1)
        prof = _lib.prof_read()
        _lib.lib.gtars_prof_enable(0)
        assert prof.get("igd_pieces_view", {"launches": 0})["launches"] == 2, prof.keys()
        assert np.array_equal(got_p, o.count_set_overlaps(qc, qs, qe, 1, n_files=F))
        assert np.array_equal(got_b, o.count_region_hits(qc, qs, qe, 1, n_files=F))
        assert np.array_equal(g.count_set_overlaps(qc, qs, qe, 3), o.count_set_overlaps(qc, qs, qe, 3, n_files=F))
        assert np.array_equal(g.count_region_hits(qc, qs, qe, 3), o.count_region_hits(qc, qs, qe, 3, n_files=F))
    sets = [big, small, _random_query_set(rng, 40_000, 2, span, 300)]
    for binary in (True, False):
        ref = o.count_region_hits if binary else o.count_set_overlaps
        assert np.array_equal(g.count_sets(sets, 1, binary), np.stack([ref(qc, qs, qe, 1, n_files=F) for qc, qs, qe in sets]))
    # the flat layout gives the same vectors (what the pieces view replaces)
    monkeypatch.setenv("GTARS_IGD_NO_PIECES", "1")
    g2 = ga.IgdIndex(c, s, e, f, np.arange(n), n_chrom=2, n_files=F)
    qc, qs, qe = small
    assert np.array_equal(g2.count_set_overlaps(qc, qs, qe, 1), o.count_set_overlaps(qc, qs, qe, 1, n_files=F))
    # ... and a pieces view that cannot be built (memory) does not fail the database: the flat layout serves it
    monkeypatch.delenv("GTARS_IGD_NO_PIECES")
    monkeypatch.setenv("GTARS_IGD_TEST_PIECES_FAIL", "1")
    g3 = ga.IgdIndex(c, s, e, f, np.arange(n), n_chrom=2, n_files=F)
    monkeypatch.delenv("GTARS_IGD_TEST_PIECES_FAIL")
    assert np.array_equal(g3.count_set_overlaps(qc, qs, qe, 1), o.count_set_overlaps(qc, qs, qe, 1, n_files=F))
    assert np.array_equal(g3.count_region_hits(qc, qs, qe, 1), o.count_region_hits(qc, qs, qe, 1, n_files=F))
    monkeypatch.setenv("GTARS_IGD_NO_PIECES", "1")  # (what follows compares with the flat index g2)
    # everything that is not a count still sees one record per stored interval
    assert len(g) == len(g2) == n  # (every record is valid here)
    ok = (qs < 2**31) & (qe < 2**31)
    assert np.array_equal(g.count_overlaps_per_query(qc[ok][:2000], qs[ok][:2000], qe[ok][:2000], 1),
                          o.count_overlaps_per_query(qc[ok][:2000], qs[ok][:2000], qe[ok][:2000], 1))


def test_igd_fine_routing_tables_boundary_probes(ga, monkeypatch):
    """The routing kernel's fine tables (IgdTiles::route_f*: ~one tile boundary per bucket, boundaries as 16-bit offsets inside
    their bucket; an offset equal to the query's is settled by the exact bound when buckets are wider than 2^16).  Queries that
    start exactly at, one before and one after EVERY ownership bound (last start of each 2048-record tile + the chromosome's
    longest record + 1) plus a random batch, with 2^16-wide buckets (exact offsets) and with 2^20-wide ones forced
    (GTARS_IGD_ROUTE_FSHIFT_MIN: 16-bp quantisation, many boundaries per bucket): per-file vectors == the oracle's, and the
    vectors of the bounds-in-global-memory form."""
    rng = np.random.default_rng(77)
    n, F, span = 300_000, 23, 60_000_000
    c = rng.integers(0, 3, n)
    s = rng.integers(0, span, n)
    e = s + rng.integers(1, 900, n)
    f = rng.integers(0, F, n)
    # the ownership bounds, restated: records in (chromosome, start) order, tiles of 2048 records inside a chromosome
    probes = []
    for ch in range(3):
        sel = c == ch
        ss = np.sort(s[sel])
        max_len = int((e[sel] - s[sel]).max())
        last = ss[2047::2048].tolist() + [int(ss[-1])]
        for b in (int(x) + max_len + 1 for x in last):
            for d in (-17, -16, -2, -1, 0, 1, 15, 16):
                if b + d >= 0:
                    probes.append((ch, b + d, b + d + 40))
    pc, ps, pe = (np.array(x, dtype=np.int64) for x in zip(*probes))
    rc, rs, re_ = _random_query_set(rng, 200_000, 3, span + 3_000, 600)
    qc = np.concatenate([pc, rc.astype(np.int64)])
    qs = np.concatenate([ps, rs.astype(np.int64)])
    qe = np.concatenate([pe, re_.astype(np.int64)])
    perm = rng.permutation(len(qc))
    qc, qs, qe = qc[perm], qs[perm], qe[perm]
    monkeypatch.setenv("GTARS_IGD_SWEEP_MIN", "1")
    vecs = []
    for fshift in (None, "20", "global"):
        if fshift == "global":
            monkeypatch.setenv("GTARS_IGD_ROUTE_BND_GLOBAL", "1")
        elif fshift:
            monkeypatch.setenv("GTARS_IGD_ROUTE_FSHIFT_MIN", fshift)
        g, o = _igd_pair(ga, c, s, e, f, np.arange(n), n_chrom=3, n_files=F)
        monkeypatch.delenv("GTARS_IGD_ROUTE_FSHIFT_MIN", raising=False)
        got = g.count_set_overlaps(qc, qs, qe, 1)
        assert np.array_equal(got, o.count_set_overlaps(qc, qs, qe, 1, n_files=F)), fshift
        assert np.array_equal(g.count_region_hits(qc, qs, qe, 1), o.count_region_hits(qc, qs, qe, 1, n_files=F)), fshift
        vecs.append(got)
    assert np.array_equal(vecs[0], vecs[1]) and np.array_equal(vecs[0], vecs[2])


def test_igd_packed_counter_flush_and_many_routing_workgroups(ga, monkeypatch):
    """Two branches only large inputs reach, forced at small sizes: (1) the sweep's 16-bit packed counters (binary counts, B16)
    are handed over in the middle of a tile when a workgroup has served 65535 queries since its last flush -- a database of
    three tiles and 400k queries, so that one workgroup serves > 100k queries of one tile, with B16 selected naturally (2000
    files, one set), forced (two sets x 1500 files) and disabled; (2) the routing kernel with MORE workgroups than CUs (batches
    beyond 16.7M queries; here the chunk is capped at 3000 queries: 400 rows of counters for the split to sum).  Per-file
    vectors == the oracle's, and the forms agree with each other."""
    rng = np.random.default_rng(31)
    n, span = 6_000, 3_000_000  # three 2048-record tiles on one chromosome
    s = rng.integers(0, span, n)
    e = s + rng.integers(1, 800, n)
    c = np.zeros(n, dtype=np.int64)
    monkeypatch.setenv("GTARS_IGD_SWEEP_MIN", "1")
    monkeypatch.setenv("GTARS_IGD_NO_HEAVY_PARTS", "1")  # (a heavy tile stays with ONE workgroup: that is the point here)
    nq = 400_000
    qc, qs, qe = _random_query_set(rng, nq, 1, span + 2_000, 500)
    for F, n_sets, env in ((2000, 1, None), (1500, 2, "GTARS_IGD_FORCE_B16"), (1500, 2, "GTARS_IGD_NO_B16")):
        f = rng.integers(0, F, n)
        g, o = _igd_pair(ga, c, s, e, f, np.arange(n), n_chrom=1, n_files=F)
        if env:
            monkeypatch.setenv(env, "1")
        if n_sets == 1:
            got = g.count_region_hits(qc, qs, qe, 1)
            assert np.array_equal(got, o.count_region_hits(qc, qs, qe, 1, n_files=F)), (F, env)
        else:
            cut = 230_000
            sets = [(qc[:cut], qs[:cut], qe[:cut]), (qc[cut:], qs[cut:], qe[cut:])]
            got = g.count_sets(sets, 1, True)
            for k, (a, b, d) in enumerate(sets):
                assert np.array_equal(got[k], o.count_region_hits(a, b, d, 1, n_files=F)), (F, env, k)
        if env:
            monkeypatch.delenv(env)
    monkeypatch.delenv("GTARS_IGD_NO_HEAVY_PARTS")
    # (2) many routing workgroups: a database with enough tiles for the two-level split (> 1024 bins), 1.2M queries in 400 chunks
    n, F, span = 2_300_000, 19, 200_000_000
    c = rng.integers(0, 2, n)
    s = rng.integers(0, span, n)
    e = s + rng.integers(1, 700, n)
    f = rng.integers(0, F, n)
    g, o = _igd_pair(ga, c, s, e, f, np.arange(n), n_chrom=2, n_files=F)
    qc, qs, qe = _random_query_set(rng, 1_200_000, 2, span + 2_000, 500)
    want = o.count_set_overlaps(qc, qs, qe, 1, n_files=F)
    assert np.array_equal(g.count_set_overlaps(qc, qs, qe, 1), want)
    monkeypatch.setenv("GTARS_IGD_ROUTE_CHUNK_MAX", "3000")
    assert np.array_equal(g.count_set_overlaps(qc, qs, qe, 1), want)
    assert np.array_equal(g.count_region_hits(qc, qs, qe, 1), o.count_region_hits(qc, qs, qe, 1, n_files=F))


def test_igd_small_database_with_more_than_16m_queries(ga):
    """A database of <= 1024 tiles takes the one-level split, whose grid was capped at 256 workgroups: beyond 16.7M queries a
    workgroup's chunk no longer fitted the routing kernel's 16-bit counters and the call failed with an internal error (round-4
    advisor finding).  17M queries (a 1M batch tiled 17 times, every copy shuffled differently) against 150k records: 17 x the
    base batch's vectors, which equal the oracle's; pairwise (rank form) and binary."""
    rng = np.random.default_rng(77)
    n, F, span = 150_000, 9, 30_000_000
    c = rng.integers(0, 2, n)
    s = rng.integers(0, span, n)
    e = s + rng.integers(1, 900, n)
    f = rng.integers(0, F, n)
    g, o = _igd_pair(ga, c, s, e, f, np.arange(n), n_chrom=2, n_files=F)
    qc, qs, qe = _random_query_set(rng, 1_000_000, 2, span + 1000, 700)
    want_p = o.count_set_overlaps(qc, qs, qe, 1, n_files=F)
    want_b = o.count_region_hits(qc, qs, qe, 1, n_files=F)
    rep = 17
    perm = np.concatenate([rng.permutation(len(qc)) + k * 0 for k in range(rep)])
    bc, bs, be = qc[perm], qs[perm], qe[perm]
    assert len(bc) > 16_800_000
    assert np.array_equal(g.count_set_overlaps(bc, bs, be, 1), rep * want_p)
    assert np.array_equal(g.count_region_hits(bc, bs, be, 1), rep * want_b)


def test_igd_rank_histogram_sweep(ga, monkeypatch):
    """Round 5: pairwise counts with min_overlap == 1 sweep without a candidate walk -- per query two ranks (among the tile's
    starts, among its SORTED ends) into two LDS histograms, per tile two prefix sums and one add per record
    (k_igd_sweep_rank; igd.rs:504-556, 753-847 is what it must reproduce).  Against the oracle's literal tile walk and against
    the walked sweep, on databases that reach every branch: chromosomes shorter than a tile, tiles cut by the chromosome's end,
    duplicate and nested records, records far longer than the halo (the per-lane global scan), equal starts and equal ends
    across tile borders, one tile that owns > 65k queries with and without heavy-tile parts (the 16-bit histograms are filled
    in rounds), batches in order and shuffled, invalid / clamped / unknown-chromosome queries."""
    monkeypatch.setenv("GTARS_IGD_SWEEP_MIN", "1")
    rng = np.random.default_rng(505)
    _lib = ga._lib

    def check(g, o, qc, qs, qe, F, label):
        want = o.count_set_overlaps(qc, qs, qe, 1, n_files=F)
        _lib.lib.gtars_prof_reset()
        _lib.lib.gtars_prof_enable(1)
        got = g.count_set_overlaps(qc, qs, qe, 1)
        facts = _lib.prof_read()
        _lib.lib.gtars_prof_enable(0)
        assert "igd_sweep_rank_form" in facts, (label, sorted(facts))
        assert np.array_equal(got, want), label
        monkeypatch.setenv("GTARS_IGD_NO_RANK", "1")
        assert np.array_equal(g.count_set_overlaps(qc, qs, qe, 1), want), label + " (walked)"
        monkeypatch.delenv("GTARS_IGD_NO_RANK")
        order = np.lexsort((qs.astype(np.int64), qc))  # in (chromosome, start) order: the sweep takes the batch as it lies
        assert np.array_equal(g.count_set_overlaps(qc[order], qs[order], qe[order], 1), want), label + " (in order)"

    # (1) many shapes of database: 5 chromosomes of very different sizes (one with 3 records, one with exactly 2048, one with 2049)
    sizes = [3, 2048, 2049, 30_000, 7000]
    c = np.concatenate([np.full(k, i) for i, k in enumerate(sizes)])
    n = len(c)
    span = 400_000
    s = rng.integers(0, span, n)
    w = rng.integers(1, 900, n)
    w[rng.integers(0, n, 40)] = rng.integers(20_000, 300_000, 40)  # longer than the halo's reach: the global scan
    e = s + w
    dup = rng.integers(0, n, 2000)  # duplicates: equal starts AND equal ends (ranks with ties)
    s[dup[:1000]] = s[dup[1000:]]
    e[dup[:1000]] = e[dup[1000:]]
    keep = c[dup[:1000]] == c[dup[1000:]]
    s[dup[:1000][~keep]] += 1
    e = np.maximum(e, s + 1)
    F = 37
    f = rng.integers(0, F, n)
    g, o = _igd_pair(ga, c, s, e, f, np.arange(n), n_chrom=len(sizes), n_files=F)
    nq = 150_000
    qc = rng.integers(0, len(sizes) + 1, nq)
    qc = np.where(qc >= len(sizes), UNK, qc)
    qs = rng.integers(0, span + 5000, nq).astype(np.int64)
    qe = qs + rng.integers(0, 3000, nq)
    qs[:30] = 0xFFFFFFF0  # negative as i32: clamped to 0
    qe[:30] = rng.integers(1, span, 30)
    qe[30:60] = 0  # rejected
    qe[60:90] = qs[60:90] + 200_000  # very wide queries
    monkeypatch.setenv("GTARS_IGD_NO_PIECES", "1")  # (long records would send the min_overlap == 1 counts to the pieces view)
    g, o = _igd_pair(ga, c, s, e, f, np.arange(n), n_chrom=len(sizes), n_files=F)
    check(g, o, qc, qs, qe, F, "shapes")
    monkeypatch.delenv("GTARS_IGD_NO_PIECES")
    # (2) equal starts / equal ends in long runs across tile borders (5000 records share 3 starts and 4 ends)
    n = 9000
    c = np.zeros(n, dtype=np.int64)
    s = np.sort(rng.choice([1000, 1000, 5000, 9000], n))
    e = s + rng.choice([10, 4000, 4000, 9000], n)
    f = rng.integers(0, 5, n)
    g, o = _igd_pair(ga, c, s, e, f, np.arange(n), n_chrom=1, n_files=5)
    qc = np.zeros(70_000, dtype=np.int64)
    qs = rng.integers(0, 20_000, 70_000).astype(np.int64)
    qe = qs + rng.integers(1, 6000, 70_000)
    check(g, o, qc, qs, qe, 5, "ties")
    # (3) one tile owns > 65k queries: with heavy-tile parts, and as ONE item (histograms filled in rounds of 65024 queries)
    n, span = 6_000, 3_000_000
    s = rng.integers(0, span, n)
    e = s + rng.integers(1, 800, n)
    c = np.zeros(n, dtype=np.int64)
    f = rng.integers(0, 700, n)
    g, o = _igd_pair(ga, c, s, e, f, np.arange(n), n_chrom=1, n_files=700)
    qc, qs, qe = _random_query_set(rng, 400_000, 1, span + 2_000, 500)
    qs[:150_000] = rng.integers(1_000_000, 1_050_000, 150_000)  # 150k queries inside one 50-kb window
    qe[:150_000] = qs[:150_000] + rng.integers(1, 500, 150_000)
    check(g, o, qc, qs, qe, 700, "heavy parts")
    monkeypatch.setenv("GTARS_IGD_NO_HEAVY_PARTS", "1")
    check(g, o, qc, qs, qe, 700, "one heavy item")
    monkeypatch.delenv("GTARS_IGD_NO_HEAVY_PARTS")
    # a database without the rank tables keeps the walk
    monkeypatch.setenv("GTARS_IGD_NO_RANK_TABLES", "1")
    g2 = ga.IgdIndex(c, s, e, f, np.arange(n), n_chrom=1, n_files=700)
    monkeypatch.delenv("GTARS_IGD_NO_RANK_TABLES")
    _lib.lib.gtars_prof_reset()
    _lib.lib.gtars_prof_enable(1)
    got = g2.count_set_overlaps(qc, qs, qe, 1)
    facts = _lib.prof_read()
    _lib.lib.gtars_prof_enable(0)
    assert "igd_sweep_rank_form" not in facts
    assert np.array_equal(got, o.count_set_overlaps(qc, qs, qe, 1, n_files=700))


def test_igd_routing_with_tile_bounds_in_global_memory(ga, monkeypatch):
    """Databases beyond ~52M records route their queries with the tile bounds in global memory instead of LDS (k_igd_route<.,
    false>; up to 65534 tiles = 134M records), and beyond 36863 tiles the split is two-level whatever the batch size.  Forced
    here on a small database (GTARS_IGD_ROUTE_BND_GLOBAL): same vectors as the oracle, shuffled and in order, one and three
    query sets."""
    rng = np.random.default_rng(4242)
    n, F, span = 400_000, 17, 80_000_000
    c = rng.integers(0, 3, n)
    s = rng.integers(0, span, n)
    e = s + rng.integers(1, 1200, n)
    f = rng.integers(0, F, n)
    g, o = _igd_pair(ga, c, s, e, f, np.arange(n), n_chrom=3, n_files=F)
    qc, qs, qe = _random_query_set(rng, 1_100_000, 3, span + 3_000, 600)
    want_p = o.count_set_overlaps(qc, qs, qe, 1, n_files=F)
    want_b = o.count_region_hits(qc, qs, qe, 1, n_files=F)
    monkeypatch.setenv("GTARS_IGD_ROUTE_BND_GLOBAL", "1")
    assert np.array_equal(g.count_set_overlaps(qc, qs, qe, 1), want_p)
    assert np.array_equal(g.count_region_hits(qc, qs, qe, 1), want_b)
    order = np.lexsort((qs, np.where((qs >= qe) | (qc == UNK) | (qe >= 2**31), 0xFFFFFFFF, qc)))
    assert np.array_equal(g.count_set_overlaps(qc[order], qs[order], qe[order], 1), want_p)
    cut = 700_000
    sets = [(qc[:cut], qs[:cut], qe[:cut]), (qc[cut:], qs[cut:], qe[cut:])]
    got = g.count_sets(sets, 1, True)
    for k, (a, b, d) in enumerate(sets):
        assert np.array_equal(got[k], o.count_region_hits(a, b, d, 1, n_files=F)), k


def test_handles_carry_their_device_and_every_entry_point_checks_it(ga):
    """Round 6: a handle records the device it was built on (gtars_index_device / gtars_igd_device).  `*_device` entry points take
    the caller's device pointers, so a handle that lives elsewhere is GTARS_ERR_INVALID_ARG ("handle lives on device k, current
    device is j") and nothing is launched; host-buffer entry points switch to the handle's device and put the caller's back.
    The test box has one GPU: the handle's device id is forged through gtars_debug_set_handle_device (include/gtars_amd_debug.h)
    -- the device forms must refuse, the host forms must TRY to switch (device 1 does not exist here: a HIP error, not a memory
    fault, and the calling thread's current device stays 0), and with the real id back everything answers as before."""
    import torch

    from gtars_amd import _lib

    lib = _lib.lib
    rng = np.random.default_rng(3)
    n, nq, F = 50_000, 80_000, 30
    c = rng.integers(0, 3, n)
    s = rng.integers(0, 5_000_000, n)
    e = s + rng.integers(1, 2_000, n)
    f = rng.integers(0, F, n)
    g, o = _pair(ga, c, s, e, n_chrom=3)
    ig, io = _igd_pair(ga, c, s, e, f, n_chrom=3, n_files=F)
    qc = rng.integers(0, 3, nq).astype(np.uint32)
    qs = rng.integers(0, 5_000_000, nq).astype(np.uint32)
    qe = (qs + rng.integers(1, 3_000, nq)).astype(np.uint32)
    dev = torch.device("cuda", torch.cuda.current_device())
    cur = torch.cuda.current_device()
    assert lib.gtars_index_device(g._h) == cur and lib.gtars_igd_device(ig._h) == cur
    d = [torch.from_numpy(x.view(np.int32)).to(dev) for x in (qc, qs, qe)]
    off = torch.empty(nq + 1, dtype=torch.int64, device=dev)
    ids = torch.empty(16 * nq, dtype=torch.int32, device=dev)
    cnt = torch.empty(nq, dtype=torch.int32, device=dev)
    hits = torch.zeros(F, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    want_cnt = o.count_overlaps(qc, qs, qe, None)
    want_hits = io.count_set_overlaps(qc, qs, qe, 1, n_files=F)

    def device_calls():
        h = g.tokenize_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, off.data_ptr(), ids.data_ptr(), ids.numel(), st)
        g.count_overlaps_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, cnt.data_ptr(), None, st)
        ig.count_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, hits.data_ptr(), 1, False, st)
        torch.cuda.synchronize()
        return h

    assert device_calls() == int(want_cnt.sum())
    assert np.array_equal(cnt.cpu().numpy().view(np.uint32), want_cnt) and np.array_equal(hits.cpu().numpy(), want_hits.astype(np.int64))
    forged = cur + 1
    assert lib.gtars_debug_set_handle_device(g._h, 0, forged) == cur
    assert lib.gtars_debug_set_handle_device(ig._h, 1, forged) == cur
    try:
        assert lib.gtars_index_device(g._h) == forged and lib.gtars_igd_device(ig._h) == forged
        for call in (
            lambda: g.tokenize_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, off.data_ptr(), ids.data_ptr(), ids.numel(), st),
            lambda: g.count_overlaps_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, cnt.data_ptr(), None, st),
            lambda: g.fill_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, off.data_ptr(), ids.data_ptr(), st),
            lambda: ig.count_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, hits.data_ptr(), 1, False, st),
            lambda: ig.count_sets_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), [0, nq // 2, nq], hits.data_ptr(), 1, True, st),
        ):
            with pytest.raises(ValueError, match=f"handle lives on device {forged}, current device is {cur}"):
                call()
        # host-buffer forms select the handle's device: on this box that device does not exist
        if torch.cuda.device_count() == forged:
            for call in (lambda: ig.count_set_overlaps(qc, qs, qe, 1), lambda: ig.count_sets([(qc, qs, qe)], 1, True),
                         lambda: g.count_overlaps(qc, qs, qe, None), lambda: g.tokenize(qc, qs, qe)):
                with pytest.raises(_lib.GtarsError, match="select the handle's device"):
                    call()
            assert torch.cuda.current_device() == cur
    finally:
        assert lib.gtars_debug_set_handle_device(g._h, 0, cur) == forged
        assert lib.gtars_debug_set_handle_device(ig._h, 1, cur) == forged
    assert device_calls() == int(want_cnt.sum())
    assert np.array_equal(ig.count_set_overlaps(qc, qs, qe, 1), want_hits)
    assert np.array_equal(g.count_overlaps(qc, qs, qe, None), want_cnt)


def test_igd_counts_from_four_host_threads_on_one_handle(ga, monkeypatch):
    """One Igd handle, four host threads, all of them starting at once on a handle nobody has queried yet: the handle's lazily
    built, mutex-guarded members (pme_file for the binary sweep, the contig tile counts for min_overlap <= 0, the host mirror,
    the values-unique flag) are built under contention, every thread has its own stream workspace.  Every result against the
    oracle (igd.rs:504-590: counts are pure functions of the database and the batch)."""
    import threading

    monkeypatch.setenv("GTARS_IGD_SWEEP_MIN", "1")
    rng = np.random.default_rng(41)
    n, F, n_chrom, span = 400_000, 50, 3, 40_000_000
    c = rng.integers(0, n_chrom, n)
    s = rng.integers(0, span, n)
    e = s + rng.integers(1, 3_000, n)
    f = rng.integers(0, F, n)
    g, o = _igd_pair(ga, c, s, e, f, np.arange(n), n_chrom=n_chrom, n_files=F)
    sets = [_random_query_set(rng, m, n_chrom, span + 5_000, 800) for m in (150_000, 90_000, 40_000, 70_000)]
    want = {}
    for k, (qc, qs, qe) in enumerate(sets):
        want[k] = (o.count_set_overlaps(qc, qs, qe, 1, n_files=F), o.count_region_hits(qc, qs, qe, 1, n_files=F),
                   o.count_region_hits(qc[:2000], qs[:2000], qe[:2000], 0, n_files=F),
                   o.count_overlaps_per_query(qc[:5000], qs[:5000], qe[:5000], 1))
    want_sets = np.stack([o.count_region_hits(qc, qs, qe, 1, n_files=F) for qc, qs, qe in sets])
    # the same calls one after the other on a handle of their own first: what fails here is not a threading matter
    g1 = ga.IgdIndex(c, s, e, f, np.arange(n), n_chrom=n_chrom, n_files=F)
    for k, (qc, qs, qe) in enumerate(sets):
        assert np.array_equal(g1.count_set_overlaps(qc, qs, qe, 1), want[k][0]), (k, "pairwise, sequential")
        assert np.array_equal(g1.count_region_hits(qc, qs, qe, 1), want[k][1]), (k, "binary, sequential")
        assert np.array_equal(g1.count_region_hits(qc[:2000], qs[:2000], qe[:2000], 0), want[k][2]), (k, "min_overlap 0, sequential")
        got = g1.count_overlaps_per_query(qc[:5000], qs[:5000], qe[:5000], 1)
        assert np.array_equal(got, want[k][3]), (k, "per query, sequential", np.argwhere(got != want[k][3])[:5].tolist())
    assert np.array_equal(g1.count_sets(sets, 1, True), want_sets)
    del g1
    errors, start = [], threading.Barrier(4)

    def body(k):
        try:
            qc, qs, qe = sets[k]
            start.wait()
            for rep in range(3):
                order = [(k + rep + j) % 5 for j in range(5)]  # every thread starts with a different call
                for what in order:
                    if what == 0:
                        assert np.array_equal(g.count_set_overlaps(qc, qs, qe, 1), want[k][0]), (k, "pairwise")
                    elif what == 1:
                        assert np.array_equal(g.count_region_hits(qc, qs, qe, 1), want[k][1]), (k, "binary")
                    elif what == 2:
                        assert np.array_equal(g.count_region_hits(qc[:2000], qs[:2000], qe[:2000], 0), want[k][2]), (k, "min_overlap 0")
                    elif what == 3:
                        assert np.array_equal(g.count_overlaps_per_query(qc[:5000], qs[:5000], qe[:5000], 1), want[k][3]), (k, "per query")
                    else:
                        assert np.array_equal(g.count_sets(sets, 1, True), want_sets), (k, "sets")
        except BaseException as ex:  # noqa: BLE001 -- reported by the main thread
            errors.append(repr(ex))

    th = [threading.Thread(target=body, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    assert g.total_records() == o.total_records()


def _sorted_batch(rng, n_chrom, nq, span, wmax, unknown=0.0):
    """queries in (chromosome id, start) order, as a sorted BED file / file-loaded RegionSet delivers them"""
    qc = rng.integers(0, n_chrom, nq).astype(np.uint32)
    qs = rng.integers(0, span, nq).astype(np.uint32)
    qe = (qs + rng.integers(0, wmax, nq)).astype(np.uint32)  # (zero-length queries included)
    if unknown:
        qc = np.where(rng.random(nq) < unknown, UNK, qc).astype(np.uint32)
    o = np.lexsort((qs, qc))
    return qc[o], qs[o], qe[o]


def _tok_device(ga, g, qc, qs, qe, hint, cap_factor=6):
    import torch

    dev = torch.device("cuda", torch.cuda.current_device())
    nq = len(qc)
    d = [torch.from_numpy(np.ascontiguousarray(x).view(np.int32)).to(dev) for x in (qc, qs, qe)]
    off = torch.empty(nq + 1, dtype=torch.int64, device=dev)
    ids = torch.empty(max(cap_factor * nq, 1024), dtype=torch.int32, device=dev)
    h = g.tokenize_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, off.data_ptr(), ids.data_ptr(), ids.numel(),
                          torch.cuda.current_stream().cuda_stream, hint=hint)
    return off.cpu().numpy().view(np.uint64), ids[:h].cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("kind", BOTH)
@pytest.mark.parametrize("explicit_ids", [False, True])
def test_sweep_tokenizer_on_batches_in_order(ga, monkeypatch, kind, explicit_ids):
    """Round 6: GTARS_TOK_SORTED -- the sweep form of the tokenizer (k_tok_sweep) for batches in (chromosome, start) order, what a
    file-loaded RegionSet is (region_set.rs:182, 502-505).  Same offsets and ids as Bits::find / AIList::find (bits.rs:141-156,
    433-446; ailist.rs:238-263) and as the default kernel, on: disjoint and overlapping universes with position-derived and
    explicit ids, several tiles with chromosome boundaries inside tiles (two and more runs per tile), unknown chromosomes
    inside the batch, zero-length and wide queries (hits beyond the 32-interval mask: the global walk), a universe far beyond
    k_tok_lds' LDS key budget, min-overlap filters, an id buffer that is too short (offsets complete, GTARS_ERR_CAPACITY)."""
    rng = np.random.default_rng(17 + kind)
    for n, n_chrom, span, wmax, qwmax, nq in ((60_000, 5, 4_000_000, 400, 600, 70_000), (3_000, 40, 100_000, 90, 200, 30_000),
                                              (400_000, 3, 60_000_000, 300, 500, 50_000), (20_000, 2, 1_000_000, 3000, 40_000, 20_000)):
        c = np.sort(rng.integers(0, n_chrom, n)).astype(np.uint32)
        s = rng.integers(0, span, n).astype(np.uint32)
        o = np.lexsort((s, c))
        c, s = c[o], s[o]
        e = (s + rng.integers(1, wmax, n)).astype(np.uint32)
        if wmax <= 400 and kind == KIND_BITS:  # a disjoint universe: clip at the next start
            nxt = np.r_[s[1:], np.uint32(0xFFFFFFFF)]
            same = np.r_[c[1:] == c[:-1], False]
            e = np.where(same, np.minimum(e, np.maximum(nxt, s + 1)), e).astype(np.uint32)
        val = rng.permutation(n).astype(np.uint32) if explicit_ids else None
        g, o_ = _pair(ga, c, s, e, val, n_chrom=n_chrom, kind=kind)
        qc, qs, qe = _sorted_batch(rng, n_chrom, nq, span + 1000, qwmax, unknown=0.01)
        want_off, want_ids = o_.tokenize(qc, qs, qe)
        _lib = ga._lib
        _lib.lib.gtars_prof_reset()
        _lib.lib.gtars_prof_enable(1)
        off, ids = _tok_device(ga, g, qc, qs, qe, g.TOK_SORTED, cap_factor=2 + qwmax // 40)
        prof = _lib.prof_read()
        _lib.lib.gtars_prof_enable(0)
        if kind == KIND_BITS or "tok_build_sweep" in prof:  # (a nested AIList universe has no blocked structure: generic kernel)
            assert "tok_build_sweep" in prof, sorted(prof)
        assert np.array_equal(off, want_off), (n, np.argwhere(off != want_off)[:3])
        assert np.array_equal(ids, want_ids), (n, np.argwhere(ids != want_ids)[:3])
        off2, ids2 = _tok_device(ga, g, qc, qs, qe, g.TOK_AUTO, cap_factor=2 + qwmax // 40)
        assert np.array_equal(off2, want_off) and np.array_equal(ids2, want_ids)
        # the same batch SHUFFLED under the hint: still right (many runs per tile -> the global path)
        p = rng.permutation(nq)[: 6000]
        w_off, w_ids = o_.tokenize(qc[p], qs[p], qe[p])
        off3, ids3 = _tok_device(ga, g, qc[p], qs[p], qe[p], g.TOK_SORTED, cap_factor=2 + qwmax // 40)
        assert np.array_equal(off3, w_off) and np.array_equal(ids3, w_ids)
    # forced small budgets: 8 staged blocks per wave (every run beyond -> global memory), one run per wave and round (the second
    # run -> global memory), none; two rounds per tile on a batch this small
    for env, val_ in (("GTARS_TOK_SWEEP_BLOCKS", "8"), ("GTARS_TOK_SWEEP_RUNS", "1"), ("GTARS_TOK_SWEEP_RUNS", "0"),
                      ("GTARS_TOK_SWEEP_ROUNDS", "2")):
        monkeypatch.setenv(env, val_)
        off5, ids5 = _tok_device(ga, g, qc, qs, qe, g.TOK_SORTED, cap_factor=2 + qwmax // 40)
        monkeypatch.delenv(env)
        assert np.array_equal(off5, want_off) and np.array_equal(ids5, want_ids), env
    # min-overlap filter through count_overlaps is another kernel; the tokenizer's filter form through find_overlaps' values
    # is covered by the fuzzer (GTARS_TOK_SWEEP=1); here: an id buffer that is too short
    import torch

    dev = torch.device("cuda", torch.cuda.current_device())
    d = [torch.from_numpy(x.view(np.int32)).to(dev) for x in (qc, qs, qe)]
    off_t = torch.empty(len(qc) + 1, dtype=torch.int64, device=dev)
    short = torch.empty(max(len(want_ids) // 3, 1), dtype=torch.int32, device=dev)
    with pytest.raises(ga.CapacityError) as ei:
        g.tokenize_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), len(qc), off_t.data_ptr(), short.data_ptr(), short.numel(),
                          torch.cuda.current_stream().cuda_stream, hint=g.TOK_SORTED)
    assert ei.value.needed == len(want_ids)
    assert np.array_equal(off_t.cpu().numpy().view(np.uint64), want_off)
    assert np.array_equal(short.cpu().numpy().view(np.uint32), want_ids[: short.numel()])


def test_sweep_tokenizer_config2_in_order_and_large_universes(ga):
    """BASELINE config 2's batch in (chromosome, start) order through the sweep form: ids AND order == oracle at 1M queries; then
    universes of 200k and 1M regions (beyond the LDS key image of the default kernel), same batch law, in order."""
    from gtars_amd import synth

    for nu in (100_000, 200_000, 1_000_000):
        u = synth.make_universe(nu)
        q = synth.make_queries(u, 1_000_000 if nu == 100_000 else 400_000)
        o = np.lexsort((q["start"], q["chrom"]))
        qc, qs, qe = (np.ascontiguousarray(q[k][o]) for k in ("chrom", "start", "end"))
        g = ga.OverlapIndex(u["chrom"], u["start"], u["end"], n_chrom=synth.N_CHROM)
        ref = oracle.Index(u["chrom"], u["start"], u["end"], None, n_chrom=synth.N_CHROM)
        want_off, want_ids = ref.tokenize(qc, qs, qe)
        off, ids = _tok_device(ga, g, qc, qs, qe, g.TOK_SORTED | g.TOK_NARROW, cap_factor=2)
        assert np.array_equal(off, want_off) and np.array_equal(ids, want_ids), nu


def test_unit_records_for_universes_of_two_block_units(ga, monkeypatch):
    """Round 6: universes beyond the LDS key budget of the tokenizer whose units are TWO blocks (~130k-260k regions) get a record
    per UNIT -- eight intervals, 64 bytes, one request per query (AccelView::rec8) -- instead of a key read that picks the block and
    then the block's 32-byte record.  Same hits in the same order as Bits::find (bits.rs:141-156, 433-446): synthetic universes of
    150k and 250k regions (both kinds of launch geometry), a small universe forced onto two-block units (GTARS_TOP_MAX) with queries
    that reach past the record's eight intervals (the tail walk from block b0 + 4), chromosome ends (sentinel slots), and the
    switch that turns the records off (GTARS_TOK_NO_UNIT_RECORDS: the round-5 path) -- which record form a launch read is taken
    from the profiling facts."""
    from gtars_amd import synth

    _lib = ga._lib

    def facts(f):
        _lib.lib.gtars_prof_reset()
        _lib.lib.gtars_prof_enable(1)
        r = f()
        p = _lib.prof_read()
        _lib.lib.gtars_prof_enable(0)
        return r, p

    for nu, nq in ((150_000, 300_000), (250_000, 2_300_000)):
        u = synth.make_universe(nu)
        q = synth.make_queries(u, nq)
        g = ga.OverlapIndex(u["chrom"], u["start"], u["end"], n_chrom=synth.N_CHROM)
        ref = oracle.Index(u["chrom"], u["start"], u["end"], None, n_chrom=synth.N_CHROM)
        want = ref.tokenize(q["chrom"], q["start"], q["end"])
        (off, ids), p = facts(lambda: _tok_device(ga, g, q["chrom"], q["start"], q["end"], g.TOK_NARROW, cap_factor=2))
        assert "tok_unit_records" in p, sorted(p)
        assert np.array_equal(off, want[0]) and np.array_equal(ids, want[1]), nu
        monkeypatch.setenv("GTARS_TOK_NO_UNIT_RECORDS", "1")
        (off2, ids2), p2 = facts(lambda: _tok_device(ga, g, q["chrom"], q["start"], q["end"], g.TOK_NARROW, cap_factor=2))
        monkeypatch.delenv("GTARS_TOK_NO_UNIT_RECORDS")
        assert "tok_block_records" in p2 and "tok_unit_records" not in p2
        assert np.array_equal(off2, want[0]) and np.array_equal(ids2, want[1]), nu
    # a small universe on two-block units, queries of up to 40 intervals, 7 chromosomes of uneven length (padding units)
    monkeypatch.setenv("GTARS_TOP_MAX", "512")
    rng = np.random.default_rng(8)
    n = 1_500
    c = np.sort(rng.choice(7, n, p=[.3, .25, .2, .1, .08, .05, .02])).astype(np.uint32)
    s = rng.integers(0, 2_000_000, n).astype(np.uint32)
    o = np.lexsort((s, c))
    c, s = c[o], s[o]
    e = (s + rng.integers(1, 3_000, n)).astype(np.uint32)
    g, o_ = _pair(ga, c, s, e, n_chrom=7)
    monkeypatch.delenv("GTARS_TOP_MAX")
    nq = 50_000
    qc = rng.integers(0, 8, nq).astype(np.uint32)
    qs = rng.integers(0, 2_010_000, nq).astype(np.uint32)
    qe = (qs + rng.integers(0, 60_000, nq)).astype(np.uint32)
    want = o_.tokenize(qc, qs, qe)
    (off, ids), p = facts(lambda: _tok_device(ga, g, qc, qs, qe, g.TOK_NARROW, cap_factor=40))
    if "tok_unit_records" in p:  # (the forced budget gave two-block units)
        assert np.array_equal(off, want[0]) and np.array_equal(ids, want[1])
    else:
        pytest.skip("GTARS_TOP_MAX=512 did not give two-block units for this universe")
