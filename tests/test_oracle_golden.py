"""Pins the CPU oracle against the reference's own known-answer tests.

Each test names the reference test (file:line) whose literals it asserts.
The reference is Rust-only and cannot run here, so these KATs + the fixture
files under tests/golden/ (copied data, see tests/golden/collect_fixtures.sh)
are what anchors the oracle -- and through it the GPU path.
"""
import gzip
import os

import numpy as np
import pytest

import oracle
from oracle import KIND_AILIST, KIND_BITS

BOTH = [KIND_AILIST, KIND_BITS]


def _ix(regions, kind, vals=None):
    """regions: list of (chrom_id, start, end)."""
    c = [r[0] for r in regions]
    s = [r[1] for r in regions]
    e = [r[2] for r in regions]
    return oracle.Index(c, s, e, vals, n_chrom=(max(c) + 1 if c else 0), kind=kind)


# ---------------------------------------------------------------- Bits / AIList

ABCD = [(0, 1, 5), (0, 3, 7), (0, 6, 10), (0, 8, 12)]  # vals a,b,c,d = 0..3


@pytest.mark.parametrize("kind", BOTH)
def test_find_overlapping_intervals(kind):
    # bits.rs:578-596 / ailist.rs:418-436
    ix = _ix(ABCD, kind)
    assert ix.chrom_len(0) == 4
    _, _, v = ix.find(0, 2, 4)
    assert set(v.tolist()) == {0, 1}
    _, _, v = ix.find(0, 9, 11)
    assert set(v.tolist()) == {2, 3}
    # bits.rs:598-605 / ailist.rs:438-445, 470-480
    assert len(ix.find(0, 13, 15)[2]) == 0
    assert len(ix.find(0, 0, 1)[2]) == 0


@pytest.mark.parametrize("kind", BOTH)
def test_empty_index(kind):
    # bits.rs:607-616 / ailist.rs:447-457
    ix = _ix([], kind)
    assert ix.chrom_len(0) == 0
    assert len(ix.find(0, 1, 2)[2]) == 0


def test_bits_doc_tests():
    # bits.rs:135-139: step-5 intervals of width 2 -> find(5,11) has 2 hits; bits.rs:331-334 count == 2
    iv = [(0, x, x + 2) for x in range(0, 100, 5)]
    ix = _ix(iv, KIND_BITS)
    assert len(ix.find(0, 5, 11)[2]) == 2
    assert ix.bits_count(0, 5, 11) == 2
    # bits.rs:194-206 (insert doc-test) restated as a build: result order (0,5,1) then (0,20,5)
    ix = oracle.Index([0, 0, 0], [0, 6, 0], [5, 10, 20], [1, 2, 5], n_chrom=1, kind=KIND_BITS)
    s, e, v = ix.find(0, 1, 3)
    assert list(zip(s.tolist(), e.tolist(), v.tolist())) == [(0, 5, 1), (0, 20, 5)]


def test_bits_order_is_start_end_then_input_order():
    # bits.rs:105 (stable sort by Interval::cmp = (start,end), interval.rs:18-31)
    ix = oracle.Index([0] * 5, [10, 10, 5, 10, 5], [30, 20, 50, 20, 50], [0, 1, 2, 3, 4], n_chrom=1, kind=KIND_BITS)
    _, _, v = ix.find(0, 0, 100)
    assert v.tolist() == [2, 4, 1, 3, 0]
    assert ix.max_len(0) == 45


AILIST_26 = [
    (0, 30), (0, 10), (0, 10), (5, 15), (5, 15), (10, 20), (10, 20), (15, 25), (15, 25), (21, 22), (22, 23),
    (20, 30), (20, 30), (25, 100), (26, 27), (27, 28), (29, 30), (30, 31), (32, 33), (50, 51), (51, 52),
    (52, 53), (53, 54), (55, 56), (60, 61), (70, 71),
]


def test_ailist_complex_interval():
    # ailist.rs:550-601
    ix = _ix([(0, s, e) for s, e in AILIST_26], KIND_AILIST)
    assert len(ix.headers(0)) == 2
    assert ix.headers(0) == [0, 24]  # second sub-list = {(0,30),(25,100)}
    s, e, _ = ix.find(0, 6, 8)
    assert len(s) == 5
    assert list(zip(s.tolist(), e.tolist())) == [(5, 15), (5, 15), (0, 10), (0, 10), (0, 30)]
    assert len(ix.find(0, 30, 35)[0]) == 3
    assert len(ix.find(0, 101, 150)[0]) == 0


def test_ailist_find_iter_matches_find_sets():
    # ailist.rs:489-519: queries (2,4),(5,8),(9,11),(0,15),(7,9) -- result sets equal Bits result sets
    a, b = _ix(ABCD, KIND_AILIST), _ix(ABCD, KIND_BITS)
    for qs, qe in [(2, 4), (5, 8), (9, 11), (0, 15), (7, 9)]:
        assert sorted(a.find(0, qs, qe)[2].tolist()) == sorted(b.find(0, qs, qe)[2].tolist())


def test_ailist_single_interval():
    # ailist.rs:521-539
    ix = _ix([(0, 5, 10)], KIND_AILIST)
    assert len(ix.find(0, 6, 8)[0]) == 1
    assert len(ix.find(0, 11, 15)[0]) == 0


# ----------------------------------------------------------- MultiChromOverlapper


@pytest.mark.parametrize("kind", BOTH)
def test_mco_basic(kind):
    # multi_chrom_overlapper.rs:715-754
    ix = _ix([(0, 100, 200), (0, 300, 400), (0, 600, 800)], kind)
    off, s, e, _ = ix.find_overlaps_regions([0], [110], [210])
    assert off.tolist() == [0, 1] and (s[0], e[0]) == (100, 200)
    # :756-792 three overlaps
    ix = _ix([(0, 100, 200), (0, 150, 250), (0, 180, 300)], kind)
    assert ix.count_overlaps([0], [160], [190]).tolist() == [3]
    # :794-823 none
    ix = _ix([(0, 100, 200), (0, 300, 400)], kind)
    assert ix.count_overlaps([0], [500], [600]).tolist() == [0]


@pytest.mark.parametrize("kind", BOTH)
def test_mco_boundary_unknown_chrom(kind):
    # :878-902 half-open boundary, :921-943 nonexistent chromosome
    ix = _ix([(0, 100, 200)], kind)
    assert ix.count_overlaps([0], [200], [300]).tolist() == [0]
    assert ix.count_overlaps([99], [100], [200]).tolist() == [0]
    assert ix.any_overlaps([99], [100], [200]).tolist() == [False]


@pytest.mark.parametrize("kind", BOTH)
def test_mco_multiple_chromosomes(kind):
    # :825-876
    ix = _ix([(0, 100, 200), (1, 300, 400), (2, 500, 600)], kind)
    assert ix.count_overlaps([0, 1], [150, 350], [250, 450]).tolist() == [1, 1]


@pytest.mark.parametrize("kind", BOTH)
def test_mco_count_any_find(kind):
    # :1070-1083
    ix = _ix([(0, 150, 200), (0, 250, 350), (0, 500, 600)], kind)
    assert ix.count_overlaps([0], [100], [300]).tolist() == [2]
    # :1085-1098
    ix = _ix([(0, 150, 250)], kind)
    assert ix.any_overlaps([0, 0], [100, 300], [200, 400]).tolist() == [True, False]
    # :1100-1116
    ix = _ix([(0, 50, 150), (0, 200, 250), (0, 400, 500)], kind)
    off, s, e, _ = ix.find_overlaps_regions([0], [100], [300])
    assert sorted(zip(s.tolist(), e.tolist())) == [(50, 150), (200, 250)]


@pytest.mark.parametrize("kind", BOTH)
def test_mco_min_overlap(kind):
    # :1118-1130
    ix = _ix([(0, 100, 110)], kind)
    assert ix.count_overlaps([0], [105], [200], min_overlap=5).tolist() == [1]
    assert ix.count_overlaps([0], [105], [200], min_overlap=6).tolist() == [0]
    assert ix.any_overlaps([0], [105], [200], min_overlap=6).tolist() == [False]


@pytest.mark.parametrize("kind", BOTH)
def test_mco_empty(kind):
    # :1132-1158
    ix = _ix([(0, 100, 200)], kind)
    assert ix.count_overlaps([], [], []).tolist() == []
    ix = _ix([], kind)
    assert ix.count_overlaps([0], [100], [200]).tolist() == [0]
    assert ix.any_overlaps([0], [100], [200]).tolist() == [False]


# -------------------------------------------------------------- IndexedRegionSet


def _irs(src, kind=KIND_AILIST):
    c, s, e = zip(*src) if src else ((), (), ())
    ix = oracle.Index(c, s, e, None, n_chrom=(max(c) + 1 if c else 0), kind=kind)
    return ix, list(c), list(s), list(e)


def _split(off, vals):
    return [vals[int(off[i]) : int(off[i + 1])].tolist() for i in range(len(off) - 1)]


def test_irs_kats():
    # indexed_region_set.rs:414-427 count == [2]
    ix, c, s, e = _irs([(0, 100, 200), (0, 150, 250), (0, 300, 400)])
    assert ix.count_overlaps([0], [180], [220]).tolist() == [2]
    # :444-459 find_overlaps == [[0,1]]
    ix, c, s, e = _irs([(0, 100, 200), (0, 300, 400)])
    off, idx = ix.irs_find_overlaps(c, s, e, [0], [150], [350])
    assert _split(off, idx) == [[0, 1]]
    # :519-539 multi chrom [1,1,0] / [T,T,F]
    ix, c, s, e = _irs([(0, 100, 200), (1, 100, 200), (2, 100, 200)])
    assert ix.count_overlaps([0, 1, 3], [150] * 3, [250] * 3).tolist() == [1, 1, 0]
    assert ix.any_overlaps([0, 1, 3], [150] * 3, [250] * 3).tolist() == [True, True, False]


def test_python_regionset_overlap_ops():
    # gtars-python/tests/test_regionset.py:37-54: a queries, b indexed (AIList default)
    a = [(0, 100, 200), (0, 300, 400), (0, 500, 600)]
    b = [(0, 150, 250), (0, 550, 650)]
    ix, c, s, e = _irs(b)
    qc, qs, qe = zip(*a)
    assert ix.count_overlaps(qc, qs, qe).tolist() == [1, 0, 1]
    assert ix.any_overlaps(qc, qs, qe).tolist() == [True, False, True]
    off, idx = ix.irs_find_overlaps(c, s, e, qc, qs, qe)
    assert _split(off, idx) == [[0], [], [1]]


def test_irs_duplicate_coordinates_return_all_rows():
    # indexed_region_set.rs:246-263 (Appendix A.11): every source row sharing a hit's coordinates, sorted+dedup
    src = [(0, 100, 200), (0, 100, 200), (0, 300, 400), (0, 100, 200)]
    ix, c, s, e = _irs(src)
    off, idx = ix.irs_find_overlaps(c, s, e, [0], [150], [160])
    assert _split(off, idx) == [[0, 1, 3]]


def test_index_side_subset_kats():
    """multi_chrom_overlapper.rs:1044-1066 (test_intersect_all, both overlapper types), :1131-1157 (empty query / index),
    indexed_region_set.rs:395-414 (test_intersect_all -> source rows 0 and 2) and :497-517."""
    src = [(0, 100, 200), (0, 300, 400), (1, 500, 600)]
    q = ([0, 1], [150, 550], [250, 650])
    for kind in (oracle.KIND_BITS, oracle.KIND_AILIST):
        c, s, e = zip(*src)
        ix = oracle.Index(c, s, e, None, n_chrom=2, kind=kind)
        oc, os_, oe = oracle.mco_subset_by_overlaps(ix, *q)
        assert list(zip(oc.tolist(), os_.tolist(), oe.tolist())) == [(0, 100, 200), (1, 500, 600)]
        assert oracle.irs_subset_by_overlaps(ix, c, s, e, *q).tolist() == [0, 2]
        assert len(oracle.mco_subset_by_overlaps(ix, [], [], [])[0]) == 0
        assert len(oracle.irs_subset_by_overlaps(ix, c, s, e, [], [], [])) == 0
    empty = oracle.Index([], [], [], None, n_chrom=1)
    assert len(oracle.mco_subset_by_overlaps(empty, [0], [100], [200])[0]) == 0
    # de-duplication (BTreeSet): the same interval hit by two queries, and two source rows with the same coordinates
    src = [(0, 100, 200), (0, 100, 200), (0, 150, 400)]
    c, s, e = zip(*src)
    ix = oracle.Index(c, s, e, None, n_chrom=1)
    oc, os_, oe = oracle.mco_subset_by_overlaps(ix, [0, 0], [120, 160], [130, 170])
    assert list(zip(os_.tolist(), oe.tolist())) == [(100, 200), (150, 400)]
    assert oracle.irs_subset_by_overlaps(ix, c, s, e, [0, 0], [120, 160], [130, 170]).tolist() == [0, 1, 2]
    # min_overlap filters only when > 1 (multi_chrom_overlapper.rs:461-465)
    oc, os_, oe = oracle.mco_subset_by_overlaps(ix, [0], [190], [260], 20)
    assert list(zip(os_.tolist(), oe.tolist())) == [(150, 400)]
    oc, os_, oe = oracle.mco_subset_by_overlaps(ix, [0], [190], [260], 1)
    assert list(zip(os_.tolist(), oe.tolist())) == [(100, 200), (150, 400)]


# ------------------------------------------------------------------- Tokenizer


def _tok(golden_dir, name):
    return oracle.OracleTokenizer(os.path.join(golden_dir, "tokenizers", name))


@pytest.mark.parametrize(
    "name",
    ["peaks.bed", "peaks.bed.gz", "tokenizer.toml", "tokenizer_ordered.toml", "tokenizer_custom_specials.toml",
     "tokenizer_ailist.toml", "tokenizer_bits.toml", "peaks.scored.bed"],
)
def test_tokenizer_vocab_size(golden_dir, name):
    # tokenizer.rs:294-333, test_tokenizers.py:44-79: 25 regions + 7 specials
    assert _tok(golden_dir, name).vocab_size == 32


def test_tokenizer_bad_type_and_custom_specials(golden_dir):
    # tokenizer.rs:336-341
    with pytest.raises(Exception):
        _tok(golden_dir, "tokenizer_bad_ttype.toml")
    # tokenizer.rs:344-360
    t = _tok(golden_dir, "tokenizer_custom_specials.toml")
    assert t.special["unk"] == "<UNKNOWN>" and t.special["pad"] == "<pad>"


@pytest.mark.parametrize("name", ["tokenizer.toml", "tokenizer_ailist.toml", "peaks.bed"])
def test_tokenize_kats(golden_dir, name):
    t = _tok(golden_dir, name)
    # tokenizer.rs:363-391 / test_tokenizers.py:103-120: no overlap / unknown chrom -> <unk> id 25
    assert t.tokenize([("chr1", 50, 150)]) == ["<unk>"]
    assert t.encode_regions([("chr1", 50, 150)]) == [25]
    assert t.encode_regions([("chr999", 50, 150)]) == [25]
    # tokenizer.rs:394-432 / test_tokenizers.py:123-139
    toks = t.tokenize([("chr1", 151399441, 151399547), ("chr2", 203871220, 203871381)])
    assert toks == ["chr1:151399431-151399527", "chr2:203871200-203871375"]
    assert [t.token_to_id(x) for x in toks] == [6, 7]


def test_tokenize_multi_overlap_order(golden_dir):
    # tokenizer.rs:465-496 / test_tokenizers.py:142-155: order [7, 8] (Bits)
    t = _tok(golden_dir, "tokenizer.toml")
    toks = t.tokenize([("chr2", 203871346, 203871616)])
    assert toks == ["chr2:203871200-203871375", "chr2:203871387-203871588"]
    assert t.encode_regions([("chr2", 203871346, 203871616)]) == [7, 8]


def test_scored_universe(golden_dir):
    # test_tokenizers.py:181-230
    t = _tok(golden_dir, "peaks.scored.bed")
    assert t.token_to_id("chr9:3526071-3526165") == 11
    assert t.universe.id_to_region[11] == "chr9:3526071-3526165"
    assert t.encode_regions([("chr9", 3526178, 3526249)]) == [10]
    # universe/mod.rs:214-228: scored file has names + scores
    assert t.universe.scores is not None and len(t.universe.scores) == 25


def test_tokenize_path_is_sorted_first(golden_dir):
    # a path is parsed AND sorted (region_set.rs:182) before tokenizing
    t = _tok(golden_dir, "peaks.bed")
    rs = oracle.read_region_set(os.path.join(golden_dir, "to_tokenize.bed"))
    assert [r[0] for r in rs] == ["chr13", "chr15", "chr15"]
    assert rs[1][1] < rs[2][1]
    assert t.encode_regions(rs) == [22, 23, 24]


def test_fragment_tokenization(golden_dir):
    # utils/fragments.rs:114-156: two barcodes from fragments1.bed.gz against consensus1.bed
    t = oracle.OracleTokenizer(os.path.join(golden_dir, "consensus", "consensus1.bed"))
    res = t.tokenize_fragment_file(os.path.join(golden_dir, "fragments", "region_scoring", "fragments1.bed.gz"))
    assert len(res) == 2
    # every non-overlapping fragment contributes exactly one unk (Appendix A.3)
    unk = t.token_to_id("<unk>")
    assert unk == 4
    assert all(len(v) >= 1 for v in res.values())


def _python_pipeline_summary(paths, om, otok):
    """the pure-Python restatement (oracle.fragsplit + tokenize_fragment_file's rules), summarised like the compiled one"""
    routed = oracle.fragsplit(os.path.dirname(paths[0]), om, file_order=[os.path.basename(p) for p in paths])
    out = {}
    for label in sorted(om.cluster_labels):
        ids, sm, bcs = 0, 0, set()
        for line in routed[label]:
            if line.startswith("#"):
                continue
            parts = line.split()
            got = otok.encode_regions([(parts[0], int(parts[1]), int(parts[2]))])
            ids, sm = ids + len(got), sm + sum(got)
            bcs.add(parts[3])
        out[label] = (ids, sm, len(bcs))
    return out


def test_compiled_fragment_pipeline_equals_the_python_restatement(golden_dir, tmp_path):
    """fragsplit_oracle.c (bench.py's CPU baseline for config 5) against oracle.fragsplit + OracleTokenizer: the reference's
    fragsplit fixtures, then files with CRLF, extra columns, '#' lines, unknown chromosomes, unmapped barcodes, multi-dot
    names and a later duplicate map key; malformed lines are errors in both."""
    import gzip

    fd = os.path.join(golden_dir, "fragments", "fragsplit")
    om = oracle.OracleBarcodeMap(os.path.join(golden_dir, "barcode_cluster_map.tsv"))
    for toml in ("tokenizer_bits.toml", "tokenizer_ailist.toml"):
        otok = oracle.OracleTokenizer(os.path.join(golden_dir, "tokenizers", toml))
        paths = sorted(os.path.join(fd, n) for n in os.listdir(fd))
        got = oracle.fragsplit_tokenize_compiled(paths, om, otok)
        assert got == _python_pipeline_summary(paths, om, otok) and sum(v[0] for v in got.values()) > 0

    rng = np.random.default_rng(5)
    ub = tmp_path / "u.bed"
    starts = np.sort(rng.integers(0, 2_000_000, 3000))
    ub.write_text("".join(f"chr{1 + i % 3}\t{s}\t{s + int(w)}\n" for i, (s, w) in enumerate(zip(starts, rng.integers(50, 4000, 3000)))))
    otok = oracle.OracleTokenizer(str(ub))
    d = tmp_path / "frags"
    d.mkdir()
    names = ["a.bed.gz", "b.sample.tsv.gz", "c.bed", "d.x.y.z.bed.gz"]
    barcodes = [f"BC{i:03d}" for i in range(40)]
    maplines = []
    for fi, name in enumerate(names):
        lines = []
        for _ in range(1500):
            s = int(rng.integers(0, 2_000_000))
            ch = ("chr1", "chr2", "chr3", "chrUn")[int(rng.integers(0, 4))]
            tail = ("\t1", "\t2\textra\tcolumns", "\t1\r")[int(rng.integers(0, 3))]
            lines.append(f"{ch}\t{s}\t{s + int(rng.integers(1, 9000))}\t{barcodes[int(rng.integers(0, 40))]}{tail}\n")
        lines.insert(7, f"#chr1\t5\t9\t{barcodes[0]}\t1\n")
        text = "".join(lines)
        if fi == 2:
            (d / name).write_text(text.rstrip("\n"))  # no trailing newline
        else:
            with gzip.open(d / name, "wt", newline="") as f:
                f.write(text)
        stem = name.split(".")[0]
        maplines += [f"{stem}+{b}\tcl{(i + fi) % 6}\n" for i, b in enumerate(barcodes) if (i + fi) % 5]
    maplines.append(f"a+{barcodes[1]}\tcl_late\n")  # the later line wins
    maplines.append("nofile+XX\tcl_empty\n")
    mp = tmp_path / "map.tsv"
    mp.write_text("".join(maplines))
    om = oracle.OracleBarcodeMap(str(mp))
    paths = [str(d / n) for n in names]
    got = oracle.fragsplit_tokenize_compiled(paths, om, otok)
    assert got == _python_pipeline_summary(paths, om, otok)
    assert got["cl_empty"] == (0, 0, 0) and got["cl_late"][2] == 1 and sum(v[0] for v in got.values()) > 3000
    (d / "bad.bed").write_text("chr1\t1\t2\tBC000\n")
    with pytest.raises(ValueError):
        oracle.fragsplit_tokenize_compiled(paths + [str(d / "bad.bed")], om, otok)
    (d / "bad.bed").write_text(f"chr1\t1\tx2\t{barcodes[2]}\t1\n")
    om.map["bad+" + barcodes[2]] = "cl0"
    with pytest.raises(ValueError):
        oracle.fragsplit_tokenize_compiled([str(d / "bad.bed")], om, otok)


def test_scoring_matrix_kat_pins_inverted_queries(golden_dir):
    # gtars-scoring/src/fragment_scoring.rs:178-206 -- [[2,2,1,3],[4,1,3,1]]; the end probe [e-5, e-6) is an
    # inverted interval, so this pins Interval::overlap (interval.rs:47-50) for inverted queries through Bits::find
    cons = oracle.read_region_set(os.path.join(golden_dir, "consensus", "consensus1.bed"))
    chrom_ids = {}
    c = [chrom_ids.setdefault(r[0], len(chrom_ids)) for r in cons]
    ix = oracle.Index(c, [r[1] for r in cons], [r[2] for r in cons], None, n_chrom=len(chrom_ids), kind=KIND_BITS)
    mat = np.zeros((2, 4), dtype=np.int64)
    for row, name in enumerate(["fragments1.bed.gz", "fragments2.bed.gz"]):
        with gzip.open(os.path.join(golden_dir, "fragments", "region_scoring", name), "rt") as f:
            for line in f:
                p = line.split()
                if not p:
                    continue
                cid = chrom_ids.get(p[0], 999)
                s, e = int(p[1]), int(p[2])
                for qs, qe in ((s + 4, s + 5), (e - 5, e - 6)):
                    for v in ix.find(cid, qs, qe)[2]:
                        mat[row, int(v)] += 1
    assert mat.tolist() == [[2, 2, 1, 3], [4, 1, 3, 1]]


# ------------------------------------------------------------------ RegionSet


def test_region_set_parse(golden_dir):
    # gtars-core/src/lib.rs:25-60: 25 regions from peaks.bed(.gz); sorted by (chr, start)
    for n in ("peaks.bed", "peaks.bed.gz"):
        rs = oracle.read_region_set(os.path.join(golden_dir, "tokenizers", n))
        assert len(rs) == 25
        keys = [(r[0].encode(), r[1]) for r in rs]
        assert keys == sorted(keys)
        assert rs[0][0] == "chr1" and rs[1][0] == "chr1"  # lexicographic: chr1 < chr10 < chr12 ...
    # header handling (region_set.rs:112-135)
    rs = oracle.read_region_set(os.path.join(golden_dir, "regionset", "dummy_headers.bed"))
    assert all(_r[1] is not None for _r in rs)


# ------------------------------------------------------------------------ IGD


def _mk_igd(recs):
    g = oracle.Igd()
    for r in recs:
        g.add(*r)
    g.finalize()
    return g


def test_igd_build_and_query_basic():
    # igd.rs:914-959
    g = _mk_igd([(0, 100, 200, 0, 0), (0, 300, 400, 0, 0), (0, 150, 250, 0, 1)])
    h = np.zeros(2, dtype=np.uint64)
    assert g.count_overlaps(0, 120, 180, 1, h) == 2 and h.tolist() == [1, 1]
    h[:] = 0
    assert g.count_overlaps(0, 350, 380, 1, h) == 1 and h.tolist() == [1, 0]
    h[:] = 0
    assert g.count_overlaps(0, 500, 600, 1, h) == 0 and h.tolist() == [0, 0]


def test_igd_min_overlap_and_tiles():
    # igd.rs:961-986
    g = _mk_igd([(0, 100, 200, 0, 0)])
    for mo, exp in ((1, 1), (10, 1), (11, 0)):
        h = np.zeros(1, dtype=np.uint64)
        g.count_overlaps(0, 190, 250, mo, h)
        assert h[0] == exp
    # igd.rs:988-1016 multi-tile spanning counted once
    g = _mk_igd([(0, 10000, 20000, 0, 0)])
    assert g.total_records() == 2
    for q in ((11000, 12000), (17000, 18000), (15000, 19000)):
        h = np.zeros(1, dtype=np.uint64)
        g.count_overlaps(0, q[0], q[1], 1, h)
        assert h[0] == 1
    # igd.rs:1018-1032 unknown chrom
    h = np.zeros(1, dtype=np.uint64)
    assert g.count_overlaps(7, 100, 200, 1, h) == 0


def test_igd_from_bed_dir(golden_dir):
    # igd.rs:1034-1057: 1 file, 3 contigs, 8 self hits
    db = oracle.OracleIgdDb.from_bed_dir(os.path.join(golden_dir, "igd_file_list_01"))
    assert len(db.file_info) == 1 and db.igd.num_contigs() == 3
    q = [("chr1", 1, 100), ("chr1", 200, 300), ("chr1", 32768, 32868), ("chr1", 49152, 49352),
         ("chr2", 1, 100), ("chr2", 200, 300), ("chr3", 32768, 32868), ("chr3", 49152, 49352)]
    assert db.count_set_overlaps(q).tolist() == [8]
    # gtars-igd/src/lib.rs:262-326: query1.bed (8 regions) -> 8 hits
    rs = oracle.read_region_set(os.path.join(golden_dir, "igd_query_files", "query1.bed"))
    assert len(rs) == 8
    assert db.count_set_overlaps(rs).tolist() == [8]


def test_igd_count_set_and_pairwise():
    # igd.rs:1161-1198
    g = _mk_igd([(0, 100, 200, 0, 0), (0, 500, 600, 0, 0), (0, 150, 250, 0, 1)])
    assert g.count_set_overlaps([0, 0], [120, 520], [180, 560], 1).tolist() == [2, 1]
    # igd.rs:1200-1221 pairwise (3) vs binary (1: enrichment.rs:830-853)
    g = _mk_igd([(0, 100, 200, 0, 0), (0, 120, 220, 0, 0), (0, 140, 240, 0, 0)])
    assert g.count_set_overlaps([0], [150], [190], 1).tolist() == [3]
    assert g.count_region_hits([0], [150], [190], 1).tolist() == [1]


def test_igd_two_set_api():
    # igd.rs:1256-1275
    g = _mk_igd([(0, 100, 200, 0, 0), (0, 300, 400, 1, 0), (0, 500, 600, 2, 0)])
    q, s = g.find_overlaps_regionset([0, 0, 0], [150, 550, 700], [350, 650, 800], 1)
    assert sorted(zip(q.tolist(), s.tolist())) == [(0, 0), (0, 1), (1, 2)]
    # igd.rs:1287-1299
    g = _mk_igd([(0, 100, 200, 0, 0)])
    assert len(g.find_overlaps_regionset([0], [190], [300], 1)[0]) == 1
    assert len(g.find_overlaps_regionset([0], [190], [300], 50)[0]) == 0
    # igd.rs:1321-1339
    g = _mk_igd([(0, 100, 200, 0, 0), (0, 150, 250, 1, 0), (0, 500, 600, 2, 0)])
    assert g.count_overlaps_per_query([0, 0, 0], [160, 550, 700], [180, 580, 800], 1).tolist() == [2, 1, 0]
    # igd.rs:1351-1369 multi-tile dedup
    g = _mk_igd([(0, 10000, 40000, 0, 0)])
    q, s = g.find_overlaps_regionset([0], [15000], [35000], 1)
    assert list(zip(q.tolist(), s.tolist())) == [(0, 0)]
    assert g.count_overlaps_per_query([0], [15000], [35000], 1).tolist() == [1]


def test_igd_negative_and_large_coordinates():
    # igd.rs:1394-1416
    g = _mk_igd([(0, -100, 200, 0, 0), (0, 100, -200, 0, 0), (0, -100, -50, 0, 0), (0, 100, 200, 0, 0)])
    h = np.zeros(1, dtype=np.uint64)
    assert g.count_overlaps(0, 150, 160, 1, h) == 1
    # igd.rs:1418-1435
    g = _mk_igd([(0, 100, 200, 0, 0)])
    assert g.count_overlaps(0, -50, 150, 1, h) == 1
    assert g.count_overlaps(0, -100, -50, 1, h) == 0
    # igd.rs:1449-1461
    g = _mk_igd([(0, 400_000_000, 400_001_000, 0, 0)])
    assert g.count_overlaps(0, 400_000_500, 400_000_600, 1, h) == 1


def test_igd_parse_bed_line():
    # igd.rs:1437-1447
    assert oracle.igd_parse_bed_line("1\t100\t200\tname\t500") == ("1", 100, 200, 500)
    assert oracle.igd_parse_bed_line("chr1\t100\t200") == ("chr1", 100, 200, -1)
    assert oracle.igd_parse_bed_line("chr1\t100\t0") is None


def test_igd_flat_formula_equals_tile_walk():
    # for min_overlap >= 1 the tile walk counts each stored interval exactly once
    rng = np.random.default_rng(7)
    n, nq, F = 3000, 400, 7
    s = rng.integers(0, 200_000, n)
    w = rng.integers(1, 40_000, n)
    e = s + w
    f = rng.integers(0, F, n)
    g = oracle.Igd()
    g.add_arrays(np.zeros(n, dtype=int), s, e, np.arange(n), f)
    g.finalize()
    qs = rng.integers(0, 220_000, nq)
    qe = qs + rng.integers(1, 50_000, nq)
    for mo in (1, 2, 10, 1000):
        got = g.count_set_overlaps(np.zeros(nq, dtype=int), qs, qe, mo, n_files=F)
        ov = np.minimum(e[None, :], qe[:, None]) - np.maximum(s[None, :], qs[:, None])
        exp = np.zeros(F, dtype=np.uint64)
        np.add.at(exp, np.broadcast_to(f[None, :], ov.shape)[ov >= mo], 1)
        assert got.tolist() == exp.tolist()
        binary = g.count_region_hits(np.zeros(nq, dtype=int), qs, qe, mo, n_files=F)
        exp_b = [(int(((ov >= mo) & (f[None, :] == k)).any(axis=1).sum())) for k in range(F)]
        assert binary.tolist() == exp_b


# ----------------------------------------------------------------------- LOLA


def test_lola_contingency_kat():
    # gtars-lola/src/enrichment.rs:879-923: a,b,c,d = 1,1,2,6
    a, b, c, d = oracle.lola_contingency([1], [2], 3, 10)
    assert (a[0], b[0], c[0], d[0]) == (1, 1, 2, 6)
    # negative cells pass through (enrichment.rs:214-235)
    a, b, c, d = oracle.lola_contingency([3], [2], 3, 4)
    assert b[0] == -1


# ----------------------------------------------------------------------- gtok


def test_gtok_golden_bytes(golden_dir, tmp_path):
    # peaks.gtok = GTOK\x01 + u16 0..24 ; tokens.gtok = (42,101,999)
    assert oracle.read_tokens_from_gtok(os.path.join(golden_dir, "out", "peaks.gtok")) == list(range(25))
    assert oracle.read_tokens_from_gtok(os.path.join(golden_dir, "out", "tokens.gtok")) == [42, 101, 999]
    p = str(tmp_path / "x" / "t.gtok")
    oracle.write_tokens_to_gtok(p, list(range(25)))
    with open(p, "rb") as f, open(os.path.join(golden_dir, "out", "peaks.gtok"), "rb") as g:
        assert f.read() == g.read()
    oracle.write_tokens_to_gtok(p, [1, 70000])
    assert open(p, "rb").read()[4] == 0x02
    assert oracle.read_tokens_from_gtok(p) == [1, 70000]


def test_splitmix64_matches_c():
    import ctypes

    st = ctypes.c_uint64(12345)
    py = oracle.SplitMix64(12345)
    for _ in range(5):
        assert oracle.lib().orc_splitmix64(ctypes.byref(st)) == py.next()


# ------------------------------------------------------------ Bits::insert / Bits::seek (doc tests of bits.rs)


def test_bits_insert_doc_example():
    # bits.rs:193-206: insert {0,20,5} into [{0,5,1},{6,10,2}]; find_iter(1,3) -> {0,5,1}, {0,20,5}
    b = oracle.MutableBits([(0, 5, 1), (6, 10, 2)])
    b.insert(0, 20, 5)
    assert len(b) == 3
    assert b.find(1, 3) == [(0, 5, 1), (0, 20, 5)]
    assert b.max_len == 20 and b.starts == [0, 0, 6] and b.ends == [5, 10, 20]


def test_bits_seek_doc_example():
    # bits.rs:351-361: intervals x..x+2 for x in 0,5,..,95; seeking each interval with one running cursor finds exactly it
    b = oracle.MutableBits([(x, x + 2, True) for x in range(0, 100, 5)])
    cursor = 0
    for s, e, _ in list(b.intervals):
        hits, cursor = b.seek(s, e, cursor)
        assert len(hits) == 1
    # bits.rs:44-58: sorted queries i..i+5
    cursor = 0
    for i in range(0, 100, 5):
        hits, cursor = b.seek(i, i + 5, cursor)
        assert hits == b.find(i, i + 5)


def test_mutable_bits_agrees_with_the_c_restatement():
    # find() of the list restatement == Index.find_overlaps_regions of oracle/gtars_oracle.c after rebuilding from the same order
    rng = np.random.default_rng(3)
    s = rng.integers(0, 2000, 300)
    e = s + rng.integers(0, 120, 300)
    b = oracle.MutableBits(list(zip(s.tolist(), e.tolist(), range(300))))
    for k in range(40):
        a = int(rng.integers(0, 2000))
        b.insert(a, a + int(rng.integers(0, 200)), 1000 + k)
    st = np.array([t[0] for t in b.intervals], dtype=np.uint32)
    en = np.array([t[1] for t in b.intervals], dtype=np.uint32)
    va = np.array([t[2] for t in b.intervals], dtype=np.uint32)
    ix = oracle.Index(np.zeros(len(st), dtype=np.uint32), st, en, va, n_chrom=1)
    assert ix.max_len(0) == b.max_len
    assert ix.stored(0)[2].tolist() == va.tolist()  # the stable build keeps the inserted order
    qs = rng.integers(0, 2100, 200).astype(np.uint32)
    qe = (qs + rng.integers(0, 150, 200)).astype(np.uint32)
    off, ids = ix.tokenize(np.zeros(200, dtype=np.uint32), qs, qe)
    for i in range(200):
        assert [t[2] for t in b.find(int(qs[i]), int(qe[i]))] == ids[off[i]:off[i + 1]].tolist()


# ------------------------------------------------------------ CLI text front ends (gtars-cli overlaprs / igd search)


def test_overlaprs_text_small(tmp_path):
    # hand-derived: chr1 has (1,5) (3,9) (8,12), chr2 has (0,4); Bits lists hits in (start, end) order, AIList -- one
    # sub-list here -- from the last start < q_end downwards; the chrX query is skipped; a line is "chr\tstart\tend"
    u = tmp_path / "u.bed"
    u.write_text("chr1\t3\t9\nchr2\t0\t4\nchr1\t1\t5\nchr1\t8\t12\textra\n")
    q = tmp_path / "q.bed"
    q.write_text("chr1\t4\t9\nchrX\t1\t2\nchr2\t3\t4\nchr1\t100\t200\n")
    assert oracle.overlaprs_text(str(u), str(q), "bits") == "chr1\t1\t5\nchr1\t3\t9\nchr1\t8\t12\nchr2\t0\t4\n"
    assert oracle.overlaprs_text(str(u), str(q), "ailist") == "chr1\t8\t12\nchr1\t3\t9\nchr1\t1\t5\nchr2\t0\t4\n"
    bad = tmp_path / "bad.bed"
    bad.write_text("chr1\t 3\t9\n")
    with pytest.raises(ValueError):
        oracle.overlaprs_text(str(bad), str(q), "bits")


def test_igd_search_text_small(tmp_path):
    a = tmp_path / "a.bed"
    a.write_text("chr1\t10\t20\nchr1\t15\t30\nchr2\t5\t6\n")
    b = tmp_path / "b.bed"
    b.write_text("chr3\t1\t2\n")
    q = tmp_path / "q.bed"
    q.write_text("chr1\t18\t19\nchr2\t0\t100\n")
    assert oracle.igd_search_text([str(a), str(b)], str(q)) == (
        "index\t number of regions\t number of hits\t File_name\n0\t3\t3\ta.bed\nTotal: 3\n")
