"""GPU tests of the reference-shaped Python surface (gtars.tokenizers / gtars.models / gtars.lola
mirrors): the reference's own python/rust test literals re-expressed, plus differential checks
against the oracle's string-level restatement."""
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tk(golden_dir):
    import gtars_amd

    assert gtars_amd.device_count() > 0
    from gtars_amd.tokenizers import Tokenizer

    return lambda name: Tokenizer(os.path.join(golden_dir, "tokenizers", name))


def R(*a):
    from gtars_amd.models import Region

    return Region(*a)


# ---- gtars-python/tests/test_tokenizers.py, same order ---------------------------------------


def test_tokenizer_initialization_variants(golden_dir):
    from gtars_amd.tokenizers import Tokenizer

    d = os.path.join(golden_dir, "tokenizers")
    assert Tokenizer.from_config(os.path.join(d, "tokenizer.toml")).vocab_size == 32
    assert Tokenizer.from_bed(os.path.join(d, "peaks.bed")).vocab_size == 32
    assert Tokenizer.from_bed(os.path.join(d, "peaks.bed.gz")).vocab_size == 32
    for n in ["peaks.bed", "peaks.bed.gz", "tokenizer.toml", "tokenizer_ordered.toml", "tokenizer_custom_specials.toml",
              "tokenizer_ailist.toml", "tokenizer_bits.toml"]:
        t = Tokenizer(os.path.join(d, n))
        assert t.vocab_size == 32 and len(t) == 32 and repr(t) == "Tokenizer(32 total regions)"
    with pytest.raises(Exception):
        Tokenizer.from_config(os.path.join(d, "tokenizer_bad_ttype.toml"))
    with pytest.raises(Exception):
        Tokenizer(os.path.join(d, "peaks.txt"))
    with pytest.raises(Exception):
        Tokenizer(os.path.join(d, "does_not_exist.bed"))


def test_tokenizer_custom_special_tokens(tk):
    t = tk("tokenizer_custom_specials.toml")
    assert t.vocab_size == 32
    assert t.unk_token == "<UNKNOWN>" and t.pad_token == "<pad>"
    assert t.special_tokens_map["unk_token"] == "<UNKNOWN>"


@pytest.mark.parametrize("cfg", ["tokenizer.toml", "tokenizer_ailist.toml", "peaks.bed"])
def test_tokenize_kats(tk, cfg):
    t = tk(cfg)
    tok = t.tokenize([R("chr1", 50, 150, None)])
    assert len(tok) == 1 and t.convert_tokens_to_ids(tok[0]) == 25
    tok = t.tokenize([R("chr999", 50, 150, None)])
    assert len(tok) == 1 and t.convert_tokens_to_ids(tok[0]) == 25
    tok = t.tokenize([R("chr1", 151399441, 151399547, None), R("chr2", 203871220, 203871381, None)])
    assert tok == ["chr1:151399431-151399527", "chr2:203871200-203871375"]
    assert [t.convert_tokens_to_ids(x) for x in tok] == [6, 7]


def test_tokenize_with_multi_overlap(tk):
    t = tk("tokenizer.toml")
    tok = t.tokenize([R("chr2", 203871346, 203871616, None)])
    assert tok == ["chr2:203871200-203871375", "chr2:203871387-203871588"]
    assert t.convert_tokens_to_ids(tok) == [7, 8]


def test_vocab_specials_encode_decode(tk):
    t = tk("peaks.scored.bed")
    vocab = t.get_vocab()
    assert len(vocab) == 32 and all(isinstance(k, str) and isinstance(v, int) for k, v in vocab.items())
    m = t.special_tokens_map
    assert isinstance(m, dict) and all(isinstance(k, str) and isinstance(v, str) for k, v in m.items())
    assert t.encode("chr9:3526071-3526165") == [11]
    assert t.encode(["chr9:3526071-3526165", "nope"]) == [11, 25]
    assert t.decode([11]) == ["chr9:3526071-3526165"] and t.decode(11) == ["chr9:3526071-3526165"]
    assert t.convert_ids_to_tokens(11) == "chr9:3526071-3526165"
    assert t.convert_ids_to_tokens([11, 9999]) == ["chr9:3526071-3526165", "<unk>"]
    assert t.get_special_tokens_mask(["<pad>", "chr9:3526071-3526165", "<unk>"]) == [1, 0, 1]
    assert (t.unk_token_id, t.pad_token_id, t.mask_token_id, t.cls_token_id, t.eos_token_id, t.bos_token_id,
            t.sep_token_id) == (25, 26, 27, 28, 29, 30, 31)


def test_tokenizer_call_and_subclass(tk, golden_dir):
    from gtars_amd.tokenizers import Tokenizer

    t = tk("peaks.scored.bed")
    enc = t([R("chr9", 3526178, 3526249, None)])
    assert enc["input_ids"] == [10] and enc["attention_mask"] == [1]
    with pytest.raises(Exception):
        enc["token_type_ids"]

    class MoreTokenizer(Tokenizer):
        def __new__(cls, *args, **kwargs):
            return super().__new__(cls, *args, **kwargs)

        def __init__(self, *args, **kwargs):
            super().__init__()

        def add_two(self, x, y):
            return x + y

        @property
        def value(self):
            return self._value

        @value.setter
        def value(self, val):
            self._value = val

    mt = MoreTokenizer(os.path.join(golden_dir, "tokenizers", "peaks.scored.bed"))
    mt.value = 5
    enc = mt([R("chr9", 3526178, 3526249, None)])
    assert enc["input_ids"] == [10] and enc["attention_mask"] == [1]
    assert mt.add_two(2, 3) == 5 and mt.value == 5


def test_tokenize_path_is_sorted_first(tk, golden_dir):
    t = tk("peaks.bed")
    p = os.path.join(golden_dir, "to_tokenize.bed")
    assert t(p)["input_ids"] == [22, 23, 24]
    with pytest.raises(FileNotFoundError):
        t.tokenize(os.path.join(golden_dir, "nope.bed"))


@pytest.mark.parametrize("cfg", ["tokenizer.toml", "tokenizer_ailist.toml", "peaks.scored.bed"])
def test_tokenizer_differential_vs_oracle(golden_dir, cfg):
    from gtars_amd.tokenizers import Tokenizer

    path = os.path.join(golden_dir, "tokenizers", cfg)
    t, o = Tokenizer(path), oracle.OracleTokenizer(path)
    assert t.get_vocab() == o.universe.region_to_id
    rng = np.random.default_rng(3)
    regs = oracle.read_region_set(os.path.join(golden_dir, "tokenizers", "peaks.bed"))
    queries = []
    for _ in range(400):
        c, s, e, _r = regs[rng.integers(0, len(regs))]
        d = int(rng.integers(-300, 300))
        w = int(rng.integers(0, 800))
        qs = max(s + d, 0)
        queries.append((c if rng.random() > 0.05 else "chrUn", qs, qs + w))
    got = t.tokenize([R(c, s, e, None) for c, s, e in queries])
    assert got == o.tokenize(queries)
    for q in queries[:40]:
        assert t.tokenize([R(*q, None)]) == o.tokenize([q])
    off, ids = t.encode_arrays([q[0] for q in queries], [q[1] for q in queries], [q[2] for q in queries])
    qc, qs, qe = o._encode_regions(queries)
    off_o, ids_o = o.index.tokenize(qc, qs, qe)
    assert off.tolist() == off_o.tolist() and ids.tolist() == ids_o.tolist()


def test_duplicate_universe_lines_quirk(tmp_path):
    """Appendix A.7: region_to_id is dense over DISTINCT strings, id_to_region has one entry per line."""
    from gtars_amd.tokenizers import Tokenizer

    p = tmp_path / "dup.bed"
    p.write_text("chr1\t100\t200\nchr1\t300\t400\nchr1\t100\t200\nchr2\t5\t50\n")
    t, o = Tokenizer(str(p)), oracle.OracleTokenizer(str(p))
    assert t.vocab_size == o.vocab_size == 3 + 7
    assert t.get_vocab() == o.universe.region_to_id
    for i in range(12):
        assert t.convert_ids_to_tokens(i) == o.universe.id_to_region.get(i, "<unk>")
    q = [("chr1", 150, 160), ("chr2", 1, 10)]
    assert t.tokenize([R(*x, None) for x in q]) == o.tokenize(q)


def test_fragment_file_tokenization(golden_dir):
    from gtars_amd.tokenizers import Tokenizer, count_fragments_by_barcode, tokenize_fragment_file

    cons = os.path.join(golden_dir, "consensus", "consensus1.bed")
    t, o = Tokenizer(cons), oracle.OracleTokenizer(cons)
    for name in ("fragments1.bed.gz", "fragments2.bed.gz"):
        f = os.path.join(golden_dir, "fragments", "region_scoring", name)
        got = tokenize_fragment_file(f, t)
        assert got == o.tokenize_fragment_file(f)
        assert len(got) == 2  # utils/fragments.rs:114-156
        cnt = count_fragments_by_barcode(f, t)
        assert {k: sum(v.values()) for k, v in cnt.items()} == {k: len(v) for k, v in got.items()}


# ---- gtars-python/tests/test_regionset.py::TestOverlapOps --------------------------------------


def _rs(*specs):
    from gtars_amd.models import RegionSet

    return RegionSet.from_regions([R(s[0], s[1], s[2], None) for s in specs])


def test_regionset_overlap_ops():
    a = _rs(("chr1", 100, 200), ("chr1", 300, 400), ("chr1", 500, 600))
    b = _rs(("chr1", 150, 250), ("chr1", 550, 650))
    assert len(a.subset_by_overlaps(b)) == 2
    assert a.count_overlaps(b) == [1, 0, 1]
    assert a.any_overlaps(b) == [True, False, True]
    assert a.find_overlaps(b) == [[0], [], [1]]
    # indexed_region_set.rs:519-539
    ref = _rs(("chr1", 100, 200), ("chr2", 100, 200), ("chr3", 100, 200))
    q = _rs(("chr1", 150, 250), ("chr2", 150, 250), ("chr4", 150, 250))
    assert q.count_overlaps(ref) == [1, 1, 0] and q.any_overlaps(ref) == [True, True, False]
    # empty sides (indexed_region_set.rs:484-516)
    empty = _rs()
    assert _rs(("chr1", 100, 200)).count_overlaps(empty) == [0]
    assert empty.count_overlaps(ref) == []


def test_regionset_files_vs_oracle(golden_dir):
    from gtars_amd.models import RegionSet

    a_path = os.path.join(golden_dir, "lola_multi_db", "collection1", "regions", "vistaEnhancers.bed")
    b_path = os.path.join(golden_dir, "lola_multi_db", "collection1", "regions", "laminB1Lads.bed")
    a, b = RegionSet(a_path), RegionSet(b_path)
    ra, rb = oracle.read_region_set(a_path), oracle.read_region_set(b_path)
    ids = {}
    bc = [ids.setdefault(r[0], len(ids)) for r in rb]
    ox = oracle.Index(bc, [r[1] for r in rb], [r[2] for r in rb], None, n_chrom=len(ids), kind=oracle.KIND_AILIST)
    qc = [ids.get(r[0], 0xFFFFFFFF) for r in ra]
    qs, qe = [r[1] for r in ra], [r[2] for r in ra]
    assert a.count_overlaps(b) == ox.count_overlaps(qc, qs, qe).tolist()
    off, idx = ox.irs_find_overlaps(bc, [r[1] for r in rb], [r[2] for r in rb], qc, qs, qe)
    assert a.find_overlaps(b) == [idx[int(off[i]):int(off[i + 1])].tolist() for i in range(len(ra))]


# ---- IGD from BED files + LOLA ------------------------------------------------------------------


def test_igd_from_bed_dir_kats(golden_dir):
    from gtars_amd.igd import Igd
    from gtars_amd.models import RegionSet

    # igd.rs:1034-1057 and gtars-igd/src/lib.rs:389-470
    g = Igd.from_bed_dir(os.path.join(golden_dir, "igd_file_list_01"))
    assert g.num_files() == 1 and g.num_contigs() == 3
    q = RegionSet(os.path.join(golden_dir, "igd_query_files", "query1.bed"))
    assert g.count_set_overlaps(q).tolist() == [8]
    hits = np.zeros(1, dtype=np.uint64)
    assert g.count_overlaps("chr1", 1, 100, 1, hits) == 1 and hits[0] == 1
    for d in ("igd_file_list_01", "igd_file_list_02"):
        g = Igd.from_bed_dir(os.path.join(golden_dir, d))
        o = oracle.OracleIgdDb.from_bed_dir(os.path.join(golden_dir, d))
        assert [(f.filename, f.num_regions) for f in g.file_info] == [(f[0], f[1]) for f in o.file_info]
        assert all(abs(f.avg_region_width - of[2]) < 1e-12 for f, of in zip(g.file_info, o.file_info))
        for qn in ("query1.bed", "query2.bed"):
            qp = os.path.join(golden_dir, "igd_query_files", qn)
            regs = oracle.read_region_set(qp)
            for mo in (1, 5):
                assert g.count_set_overlaps(RegionSet(qp), mo).tolist() == o.count_set_overlaps(regs, mo).tolist()
                assert g.count_region_hits(RegionSet(qp), mo).tolist() == o.count_region_hits(regs, mo).tolist()


def test_igd_api_kats():
    from gtars_amd.igd import FileInfo, Igd

    # igd.rs:914-959
    g = Igd()
    g.file_info = [FileInfo("file0.bed", 2, 100.0), FileInfo("file1.bed", 1, 50.0)]
    g.add("chr1", 100, 200, 0, 0)
    g.add("chr1", 300, 400, 0, 0)
    g.add("chr1", 150, 250, 0, 1)
    g.finalize()
    h = np.zeros(2, dtype=np.uint64)
    assert g.count_overlaps("chr1", 120, 180, 1, h) == 2 and h.tolist() == [1, 1]
    assert g.count_overlaps("chrZ", 120, 180, 1, h) == 0
    # igd.rs:1418-1435 negative start is clamped, both negative -> 0
    g = Igd.from_region_sets([("test.bed", [("chr1", 100, 200)])])
    h = np.zeros(1, dtype=np.uint64)
    assert g.count_overlaps("chr1", -50, 150, 1, h) == 1
    assert g.count_overlaps("chr1", -100, -50, 1, h) == 0
    # igd.rs:1371-1392 invalid intervals skipped in count / width
    g = Igd.from_region_sets([("test.bed", [("chr1", 100, 200), ("chr1", 300, 300), ("chr1", 500, 400), ("chr1", 600, 700)])])
    assert g.file_info[0].num_regions == 2 and abs(g.file_info[0].avg_region_width - 100.0) < 1e-9
    # igd.rs:1256-1275, 1321-1339
    from gtars_amd.models import RegionSet

    subject = RegionSet.from_vectors(["chr1"] * 3, [100, 300, 500], [200, 400, 600])
    query = RegionSet.from_vectors(["chr1"] * 3, [150, 550, 700], [350, 650, 800])
    g = Igd.from_single_region_set(subject)
    assert sorted(g.find_overlaps_regionset(query, 1)) == [(0, 0), (0, 1), (1, 2)]
    assert g.count_overlaps_per_query(query, 1) == [2, 1, 0]


def _write_beds(tmp_path, specs):
    paths = []
    for name, regs in specs:
        p = tmp_path / name
        p.write_text("".join(f"{c}\t{s}\t{e}\n" for c, s, e in regs))
        paths.append(str(p))
    return paths


def test_run_lola_kats(tmp_path):
    from gtars_amd.lola import RegionDB, lola_counts, run_lola

    # enrichment.rs:724-775 (basic), :879-923 (a,b,c,d = 1,1,2,6), :830-853 (binary counting)
    paths = _write_beds(tmp_path, [("db0.bed", [("chr1", 100, 200), ("chr1", 300, 400)])])
    db = RegionDB.from_bed_files(paths)
    user = [("chr1", 150, 180), ("chr1", 500, 600), ("chr1", 700, 800)]
    universe = [("chr1", 50, 250), ("chr1", 250, 450), ("chr1", 450, 550), ("chr1", 550, 650), ("chr1", 650, 750),
                ("chr1", 750, 850), ("chr1", 850, 950), ("chr1", 950, 1050), ("chr1", 1050, 1150), ("chr1", 1150, 1250)]
    res = run_lola([user], universe, db)
    assert (res["support"], res["b"], res["c"], res["d"]) == ([1], [1], [2], [6])
    assert res["filename"] == ["db0.bed"] and res["rnkPV"] == [1] and res["size"] == [2]
    assert 0.0 <= res["qValue"][0] <= 1.0

    paths = _write_beds(tmp_path, [("db3.bed", [("chr1", 100, 200), ("chr1", 120, 220), ("chr1", 140, 240)])])
    db = RegionDB.from_bed_files(paths)
    res = run_lola([[("chr1", 150, 190)]], [("chr1", 50, 300), ("chr1", 400, 500)], db)
    assert res["support"] == [1]

    # two db sets / two user sets (enrichment.rs:724-775, 796-828) + python test fixture shape
    paths = _write_beds(tmp_path, [("db1.bed", [("chr1", 100, 200), ("chr1", 500, 600)]),
                                   ("db2.bed", [("chr1", 150, 250), ("chr1", 700, 800)])])
    db = RegionDB.from_bed_files(paths)
    assert db.num_region_sets == 2 and db.list_region_sets() == ["db1.bed", "db2.bed"]
    universe = [("chr1", 50, 250), ("chr1", 450, 650), ("chr1", 650, 850), ("chr1", 900, 1000), ("chr1", 1100, 1200)]
    res = run_lola([[("chr1", 120, 180)], [("chr1", 120, 180), ("chr1", 520, 560)]], universe, db)
    assert len(res["userSet"]) == 4
    rows = {(u, f): s for u, f, s in zip(res["userSet"], res["dbSet"], res["support"])}
    assert rows == {(0, 0): 1, (0, 1): 1, (1, 0): 2, (1, 1): 1}
    assert res["pValueLog"] == sorted(res["pValueLog"], reverse=True)
    with pytest.raises(ValueError):
        run_lola([[("chr1", 1, 2)]], universe, db, direction="sideways")
    with pytest.raises(RuntimeError):
        run_lola([[("chr1", 1, 2)]], [], db)
    # the counts agree with the oracle's Igd on the same inputs
    o = oracle.OracleIgdDb(paths)
    uni_hits, user_hits, _ = lola_counts([[("chr1", 120, 180), ("chr1", 520, 560)]], universe, db)
    assert uni_hits.tolist() == o.count_region_hits(universe).tolist()
    assert user_hits[0].tolist() == o.count_region_hits([("chr1", 120, 180), ("chr1", 520, 560)]).tolist()


def test_lola_from_folder(golden_dir):
    from gtars_amd.lola import RegionDB, run_lola

    db = RegionDB.from_folder(os.path.join(golden_dir, "lola_multi_db"))
    annos = db.collection_anno
    assert isinstance(annos, list) and len(annos) >= 1
    assert set(annos[0].keys()) == {"collectionname", "collector", "date", "source", "description"}
    assert "collection1" in [a["collectionname"] for a in annos]
    assert db.num_region_sets == 6
    user = oracle.read_region_set(os.path.join(golden_dir, "lola_multi_db", "collection1", "regions", "vistaEnhancers.bed"))
    uni = user + oracle.read_region_set(os.path.join(golden_dir, "lola_multi_db", "collection1", "regions", "cpgIslandExt.bed"))
    res = run_lola([[r[:3] for r in user[:50]]], [r[:3] for r in uni], db)
    assert len(res["dbSet"]) == 6
    k = res["filename"].index("vistaEnhancers.bed")
    assert res["support"][k] == min(50, len(user))


def test_igd_save_and_reload(tmp_path, golden_dir):
    """igd.rs:1085-1124 (save/reload equality) + the on-disk layout of igd.rs:425-486."""
    import struct

    from gtars_amd.igd import FileInfo, Igd

    g = Igd()
    g.file_info = [FileInfo("file0.bed", 2, 100.0), FileInfo("file1.bed", 1, 100.0)]
    g.add("chr1", 100, 200, 5, 0)
    g.add("chr1", 10000, 40000, 7, 0)   # spans tiles 0..2 -> three replicas on disk
    g.add("chr2", 150, 250, 9, 1)
    g.finalize()
    p = str(tmp_path / "out" / "db.igd")
    g.save(p)
    raw = open(p, "rb").read()
    nbp, gtype, nctg = struct.unpack_from("<3i", raw, 0)
    assert (nbp, gtype, nctg) == (16384, 1, 2)
    ntiles = struct.unpack_from("<2i", raw, 12)
    assert ntiles == (3, 1)
    counts = struct.unpack_from("<4i", raw, 20)
    assert counts == (2, 1, 1, 1) and sum(counts) == g.total_records() == 5
    assert raw[36:40] == b"chr1" and raw[36 + 40:36 + 44] == b"chr2"
    first = struct.unpack_from("<4i", raw, 36 + 80)
    assert first == (0, 100, 200, 5)
    assert len(raw) == 36 + 80 + 16 * 5
    tsv = open(str(tmp_path / "out" / "db.tsv")).read().splitlines()
    assert tsv[0] == "Index\tFile\tNumber of Regions\tAvg size" and tsv[1] == "0\tfile0.bed\t2\t100.00"
    r = Igd.from_igd_file(p)
    assert r.num_files() == 2 and r.num_contigs() == 2 and r.total_records() == 5
    q = [("chr1", 120, 180), ("chr1", 15000, 35000), ("chr2", 100, 300), ("chr3", 1, 2)]
    assert r.count_set_overlaps(q).tolist() == g.count_set_overlaps(q).tolist() == [2, 1]
    assert r.count_region_hits(q).tolist() == g.count_region_hits(q).tolist()
    # a database built by the C++ host from BED files survives the round trip too
    d = Igd.from_bed_dir(os.path.join(golden_dir, "igd_file_list_02"))
    p2 = str(tmp_path / "two.igd")
    d.save(p2)
    d2 = Igd.from_igd_file(p2)
    o = oracle.OracleIgdDb.from_bed_dir(os.path.join(golden_dir, "igd_file_list_02"))
    assert d2.total_records() == o.igd.total_records()
    regs = oracle.read_region_set(os.path.join(golden_dir, "igd_query_files", "query1.bed"))
    assert d2.count_set_overlaps(regs).tolist() == o.count_set_overlaps(regs).tolist()
    assert [f.filename for f in d2.file_info] == [f[0] for f in o.file_info]


def test_region_scoring_matrix_kat(golden_dir):
    """gtars-scoring/src/fragment_scoring.rs:178-206: ATAC-mode matrix [[2,2,1,3],[4,1,3,1]] (2 files x 4 peaks)."""
    from gtars_amd.scoring import barcode_scoring_from_fragments, region_scoring_from_fragments

    cons = os.path.join(golden_dir, "consensus", "consensus1.bed")
    mat = region_scoring_from_fragments(os.path.join(golden_dir, "fragments", "region_scoring", "*.bed.gz"), cons, "atac")
    assert mat.shape == (2, 4) and mat.tolist() == [[2, 2, 1, 3], [4, 1, 3, 1]]
    chip = region_scoring_from_fragments(os.path.join(golden_dir, "fragments", "region_scoring", "*.bed.gz"), cons, "chip")
    # ChIP mode == per-fragment overlaps: same as the tokenizer's view of the file
    t = oracle.OracleTokenizer(cons)
    for row, name in enumerate(["fragments1.bed.gz", "fragments2.bed.gz"]):
        toks = t.tokenize_fragment_file(os.path.join(golden_dir, "fragments", "region_scoring", name))
        exp = np.zeros(4, dtype=np.int64)
        for ids in toks.values():
            for i in ids:
                if i < 4:
                    exp[i] += 1
        assert chip[row].tolist() == exp.tolist()
    bc = barcode_scoring_from_fragments(os.path.join(golden_dir, "fragments", "region_scoring", "fragments1.bed.gz"), cons)
    assert sum(sum(d.values()) for d in bc.values()) == int(chip[0].sum())
    with pytest.raises(ValueError):
        region_scoring_from_fragments([], cons, "nope")


def test_barcode_scoring_on_the_device_and_matrix_market_output(tmp_path):
    """barcode_scoring_from_fragments (fragment_scoring.rs:125-155) counts (barcode, peak) pairs in a device-resident band of the
    barcode x peak matrix -- one band and many small ones give the oracle's per-barcode peak counts -- and
    write_sparse_counts_to_mtx (matrix_market.rs:26-92) writes what scipy reads back as that matrix."""
    import gzip

    from scipy.io import mmread

    from gtars_amd import scoring, synth

    u = synth.make_universe(3_000)
    ub, fd, _mp = synth.write_config5_inputs(str(tmp_path), u, 1, 20_000, 5, barcodes=300)[:3]
    frag = os.path.join(fd, sorted(os.listdir(fd))[0])
    # a consensus set's peak index is the region's rank in (chr string, start) order (a file-loaded RegionSet is sorted,
    # region_set.rs:182), a tokenizer's id its line number: write the universe in that order so that the two coincide
    rows = sorted(zip((synth.CHROM_NAMES[c] for c in u["chrom"]), u["start"].tolist(), u["end"].tolist()))
    ub = str(tmp_path / "consensus_sorted.bed")
    with open(ub, "w") as fh:
        fh.write("".join(f"{c}\t{a}\t{b}\n" for c, a, b in rows))
    t = oracle.OracleTokenizer(ub)
    n_peaks = len(u["chrom"])
    exp = {}
    for bcode, ids in t.tokenize_fragment_file(frag).items():
        row = {}
        for i in ids:
            if i < n_peaks:  # (a fragment without a hit contributes the unk id: not a peak)
                row[i] = row.get(i, 0) + 1
        if row:
            exp[bcode] = row
    band0 = scoring.BAND_CELLS
    try:
        for cells in (band0, 7 * n_peaks, n_peaks):  # all barcodes at once; 7 per band; 1 per band
            scoring.BAND_CELLS = cells
            got = scoring.barcode_scoring_from_fragments(frag, ub)
            assert got == exp, cells
    finally:
        scoring.BAND_CELLS = band0
    prefix = str(tmp_path / "out")
    scoring.write_sparse_counts_to_mtx(got, n_peaks, prefix)
    m = mmread(gzip.open(prefix + "_matrix.mtx.gz")).tocsr()
    names = gzip.open(prefix + "_barcodes.tsv.gz", "rt").read().split()
    assert names == sorted(exp) and m.shape == (len(exp), n_peaks) and m.nnz == sum(len(r) for r in exp.values())
    for ri, bcode in enumerate(names):
        assert {int(k): int(v) for k, v in zip(m[ri].indices, m[ri].data)} == exp[bcode]
    assert gzip.open(prefix + "_features.tsv.gz", "rt").read().split()[:2] == ["peak_0", "peak_1"]


def test_fragment_files_in_parallel_host_threads(tmp_path):
    """tokenize_fragment_files: many files through a host thread pool (one device workspace per thread, the
    index shared read-only) give exactly the per-file results, in input order."""
    import gzip

    from gtars_amd import synth
    from gtars_amd.tokenizers import Tokenizer, tokenize_fragment_file, tokenize_fragment_files

    u = synth.make_universe(20_000)
    ub = tmp_path / "u.bed"
    ub.write_text("".join(f"{synth.CHROM_NAMES[c]}\t{s}\t{e}\n" for c, s, e in zip(u["chrom"], u["start"], u["end"])))
    tok = Tokenizer.from_bed(str(ub))
    files = []
    for k in range(12):
        q = synth.make_queries(u, 20_000, seed=50 + k)
        rng = np.random.default_rng(k)
        text = "".join(f"{synth.CHROM_NAMES[c] if c < synth.N_CHROM else 'chrUn_x'}\t{s}\t{e}\tBC{b:03d}\t1\n"
                       for c, s, e, b in zip(q["chrom"], q["start"], q["end"], rng.integers(0, 50, len(q["chrom"]))))
        p = tmp_path / f"f{k}.tsv.gz"
        with gzip.open(p, "wt") as fh:
            fh.write(text)
        files.append(str(p))
    serial = [tokenize_fragment_file(f, tok) for f in files]
    for rep in range(3):
        assert tokenize_fragment_files(files, tok, workers=8) == serial
    # and against the oracle, every file
    o = oracle.OracleTokenizer(str(ub))
    for f, got in zip(files, serial):
        assert got == o.tokenize_fragment_file(f), f


def test_bench_line_contract():
    """bench.py prints ONE JSON line with the keys the driver and the judge read."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["higher_is_better"] is True and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source"):
        assert k in r, k
    # the two child `rocprofv3 --pmc` passes ran (or the committed summary was replayed): HBM-side bytes per launch,
    # never below the 12 B per query the kernel must read
    assert r["traffic"] is not None and r["traffic"] > 12.0 * d["config"]["queries_per_step_per_gpu"], r
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert d["value"] > 1e9 and abs(d["value"] - d["config"]["queries_per_step_per_gpu"] / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]


# ------------------------------------------------------------ two ranks on one GPU: the sharded drivers on the HIP path


def _hip_rank(rank, world, port, q):
    import os

    import torch
    import torch.distributed as dist

    from gtars_amd import sharding
    from test_sharding_gloo import sharded_checks

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, sharded_checks(sharding.HipEngine("cuda:0"), rank, world)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_one_gpu_sharded_drivers():
    """world size 2 on ONE MI355X (gloo carries the collectives; under RCCL the same tensors are reduced in place):
    chromosome-bucket == range == single-process for the IGD / LOLA vectors, and the gathered CSR == the unsharded one.
    Same assertions as the CPU gloo test, with HipEngine behind the drivers."""
    import sys

    import torch.multiprocessing as mp

    from test_sharding_gloo import _free_port

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hip_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for rank, checks in res.items():
        assert checks and all(checks.values()), (rank, checks)
    assert res[0]["bucket_db_is_cut"]


def _rccl_world1_rank(port, q):
    """ONE rank, backend nccl (= RCCL): the drivers' collectives run on device tensors, no host staging."""
    import os

    import numpy as np
    import torch
    import torch.distributed as dist

    from gtars_amd import sharding, synth
    from test_sharding_gloo import sharded_checks

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        out = {"backend_is_nccl": dist.get_backend() == "nccl"}
        eng = sharding.HipEngine(dev)
        # the direct calls the verdict names, against the unsharded vectors
        F = 41
        db = synth.make_igd_db(150_000, F)
        bq = synth.make_background_queries(120_000)
        whole = eng.igd(db["chrom"], db["start"], db["end"], db["file"], synth.N_CHROM, F)
        sdb = sharding.ShardedIgd(eng, db, synth.N_CHROM, F, mode="bucket", balance_with=[bq["chrom"]])
        out["collective_enabled_at_world_1"] = bool(sdb.collective and sdb.world == 1)
        h = sdb.upload_local(bq)
        for binary in (False, True):
            exp = eng.igd_count(whole, bq["chrom"], bq["start"], bq["end"], 1, binary)
            got = sdb.count_resident(h, 1, binary)  # count + dist.all_reduce(hits) on the device tensor
            torch.cuda.synchronize()
            out[f"count_resident_{'binary' if binary else 'pairwise'}"] = bool(got.is_cuda and torch.equal(got, exp))
        uni = synth.make_universe(60_000, seed=3)
        sel = np.sort(np.random.default_rng(2).choice(len(uni["chrom"]), 9_000, replace=False))
        user = {k: uni[k][sel] for k in ("chrom", "start", "end")}
        hs = sdb.upload_local_sets([uni, user])
        got = sdb.count_sets_resident(hs, 1, True)
        torch.cuda.synchronize()
        exp = torch.stack([eng.igd_count(whole, x["chrom"], x["start"], x["end"], 1, True) for x in (uni, user)])
        out["count_sets_resident"] = bool(got.is_cuda and torch.equal(got, exp))
        u = synth.make_universe(5_000)
        tq = synth.make_queries(u, 50_001)
        ix = eng.index(u["chrom"], u["start"], u["end"], synth.N_CHROM)
        off, ids = eng.tokenize(ix, tq["chrom"], tq["start"], tq["end"])
        g_off, g_ids, plan = sharding.all_gather_csr_device(off, ids, int(ids.numel()), return_plan=True)
        g2_off, g2_ids = sharding.all_gather_csr_device(off, ids, int(ids.numel()), plan=plan)
        torch.cuda.synchronize()
        out["all_gather_csr_device"] = bool(g_off.is_cuda and torch.equal(g_off, off) and torch.equal(g_ids, ids)
                                            and torch.equal(g2_off, off) and torch.equal(g2_ids, ids))
        # and the whole driver checklist of the gloo tests under RCCL
        out.update({"drivers_" + k: v for k, v in sharded_checks(eng, 0, 1).items()})
        q.put(out)
    finally:
        dist.destroy_process_group()


def test_rccl_world_size_one_executes_the_device_collectives():
    """The RCCL branch of the collectives (all_reduce of the per-file vector(s), all_gather_into_tensor of the CSR) on DEVICE
    tensors: a world-size-1 `nccl` process group on the one GPU of the test box initialises an RCCL communicator and runs the
    same calls an 8-GPU job makes (what is reduced: gtars-lola/src/enrichment.rs:198-221).  Results == the unsharded ones."""
    import sys

    import torch.multiprocessing as mp

    from test_sharding_gloo import _free_port

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_world1_rank, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=60)
    assert res and all(res.values()), res
    assert "drivers_tokenize_gather_plan_reuse" in res and "count_sets_resident" in res


def _hip_fragment_rank(rank, world, port, q, ub, fd, mp_path):
    import os

    import torch
    import torch.distributed as dist

    import oracle
    from gtars_amd import fragsplit
    from gtars_amd.fragsplit import BarcodeToClusterMap
    from gtars_amd.tokenizers import Tokenizer
    from test_sharding_gloo import fragment_pipeline_checks, oracle_fragment_pipeline, same_cluster_results

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tok, m = Tokenizer.from_bed(ub), BarcodeToClusterMap.from_file(mp_path)
        expected = fragsplit.fragsplit_tokenize(fd, m, tok, as_arrays=True)  # the single-process product result ...
        out = fragment_pipeline_checks(fd, m, tok, None, expected, world)
        if rank == 0:  # ... which is the oracle's
            out["single_process_equals_oracle"] = same_cluster_results(
                expected, oracle_fragment_pipeline(fragsplit.list_fragment_files(fd), oracle.OracleBarcodeMap(mp_path), oracle.OracleTokenizer(ub)))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_one_gpu_fragment_pipeline(tmp_path):
    """BASELINE config 5's multi-GPU split on the product path: two ranks (gloo, sharing this GPU) each run the
    fused fragsplit -> tokenizer pipeline on their run of the sorted file list; gathered and rank-local + merged results == the
    single-process pipeline == the oracle (split.rs:36-151, utils/fragments.rs:61-82)."""
    import sys

    import torch.multiprocessing as mp

    from test_sharding_gloo import _free_port, write_fragment_inputs

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    ub, fd, mpth = write_fragment_inputs(tmp_path, files=13, frags=1500, clusters=5)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hip_fragment_rank, args=(r, 2, port, q, ub, fd, mpth)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [0, 1]
    for rank, checks in res.items():
        assert checks and all(checks.values()), (rank, checks)
    assert res[0]["single_process_equals_oracle"]


def test_bench_tools_run_with_two_ranks_on_one_gpu():
    """tools/igd_bench.py and tools/lola_bench.py --gpus 2 under torch.distributed.run (gloo: both ranks on this GPU) give the
    same totals as their single-GPU runs (small sizes)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GTARS_BENCH_BACKEND="gloo", NDB="300000", NQ="100000", F="50", PER="1000", NUNI="20000", NUSER="3000")

    def run(tool, n):
        cmd = [sys.executable]
        if n > 1:
            from test_sharding_gloo import _free_port

            cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                    "--master-port", str(_free_port())]
        cmd += [os.path.join(root, "tools", tool), "--gpus", str(n)]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert r.returncode == 0 and lines, (r.stdout[-2000:], r.stderr[-2000:])
        return json.loads(lines[-1])

    i1, i2 = run("igd_bench.py", 1), run("igd_bench.py", 2)
    for k in ("pairwise", "binary", "pairwise_sorted_input", "binary_sorted_input"):
        assert i1[k]["total_hits"] == i2[k]["total_hits"], k
    assert i2["n_gpus"] == 2 and i2["local_db_intervals"] < i1["local_db_intervals"]
    l1, l2 = run("lola_bench.py", 1), run("lola_bench.py", 2)
    assert l1["identities_hold"] and l2["identities_hold"] and l1["support_sum"] == l2["support_sum"]


# ------------------------------------------------------------ the host pipeline (gtars_tokenize_into)


def test_tokenize_into_streaming_pipeline_matches_oracle(monkeypatch):
    """gtars_tokenize_into: chunked H2D / kernel / D2H pipeline with chained launches (every chunk starts its offsets at
    the running total of the chunks before).  Bit-exact offsets and ids for 1..16 chunks, reused output buffers, a too
    small ids buffer, an AIList-order index (generic kernel: one chunk) and calls from several host threads."""
    from concurrent.futures import ThreadPoolExecutor

    import gtars_amd
    from gtars_amd import synth

    u = synth.make_universe(40_000)
    q = synth.make_queries(u, 1_300_001)
    ix = gtars_amd.OverlapIndex(u["chrom"], u["start"], u["end"], n_chrom=synth.N_CHROM)
    ref = oracle.Index(u["chrom"], u["start"], u["end"], None, n_chrom=synth.N_CHROM)
    off_o, ids_o = ref.tokenize(q["chrom"], q["start"], q["end"])
    out = (np.empty(len(q["chrom"]) + 1, dtype=np.uint64), np.empty(len(ids_o) + 100, dtype=np.uint32))
    for chunks in ("", "1", "3", "16"):
        if chunks:
            monkeypatch.setenv("GTARS_PIPE_CHUNKS", chunks)
        out[0][:] = 0xFFFFFFFF
        out[1][:] = 0xFFFFFFFF
        off, ids = ix.tokenize(q["chrom"], q["start"], q["end"], out=out)
        assert np.array_equal(off, off_o) and np.array_equal(ids, ids_o), chunks
        assert ids.base is out[1] or ids is out[1]  # a view of the caller's buffer
    monkeypatch.delenv("GTARS_PIPE_CHUNKS")
    # ids buffer too small: the binding retries with the size the library reports; a prefix of the batch needs less
    small = (out[0], np.empty(10, dtype=np.uint32))
    off, ids = ix.tokenize(q["chrom"], q["start"], q["end"], out=small)
    assert np.array_equal(off, off_o) and np.array_equal(ids, ids_o)
    n = 70_000
    off_p, ids_p = ref.tokenize(q["chrom"][:n], q["start"][:n], q["end"][:n])
    off, ids = ix.tokenize(q["chrom"][:n], q["start"][:n], q["end"][:n], out=out)
    assert np.array_equal(off, off_p) and np.array_equal(ids, ids_p)
    off, ids = ix.tokenize(q["chrom"][:0], q["start"][:0], q["end"][:0], out=out)
    assert off.tolist() == [0] and len(ids) == 0
    # the allocating form runs on the same pipeline
    off, ids = ix.tokenize(q["chrom"], q["start"], q["end"])
    assert np.array_equal(off, off_o) and np.array_equal(ids, ids_o)
    # AIList order: generic kernel, single chunk
    ia = gtars_amd.OverlapIndex(u["chrom"], u["start"], u["end"], n_chrom=synth.N_CHROM, kind=oracle.KIND_AILIST)
    ra = oracle.Index(u["chrom"], u["start"], u["end"], None, n_chrom=synth.N_CHROM, kind=oracle.KIND_AILIST)
    off_a, ids_a = ra.tokenize(q["chrom"][:n], q["start"][:n], q["end"][:n])
    off, ids = ia.tokenize(q["chrom"][:n], q["start"][:n], q["end"][:n], out=out)
    assert np.array_equal(off, off_a) and np.array_equal(ids, ids_a)

    # every host thread has its own pipeline (device buffers, streams, helper thread)
    def work(t):
        lo = t * 100_000
        sl = slice(lo, lo + 300_000)
        mine = (np.empty(300_001, dtype=np.uint64), np.empty(400_000, dtype=np.uint32))
        o1, i1 = ix.tokenize(q["chrom"][sl], q["start"][sl], q["end"][sl], out=mine)
        o2, i2 = ref.tokenize(q["chrom"][sl], q["start"][sl], q["end"][sl])
        return bool(np.array_equal(o1, o2) and np.array_equal(i1, i2))

    with ThreadPoolExecutor(max_workers=4) as ex:
        assert all(ex.map(work, range(8)))


# ------------------------------------------------------------ fragsplit -> tokenizer (BASELINE config 5)


def test_fragsplit_tokenize_pipeline_equals_two_step(tk, golden_dir, tmp_path):
    """The fused pipeline returns, per cluster, exactly what tokenizing the written cluster file returns -- which in turn
    equals the oracle's tokenize_fragment_file of the oracle's fragsplit output (reference fixtures + a larger synthetic set)."""
    import gzip

    from gtars_amd.fragsplit import BarcodeToClusterMap, fragsplit_tokenize, pseudobulk_fragment_files
    from gtars_amd.tokenizers import tokenize_fragment_file

    otk = oracle.OracleTokenizer(os.path.join(golden_dir, "tokenizers", "peaks.bed"))
    tk = tk("peaks.bed")
    cases = [(os.path.join(golden_dir, "fragments", "fragsplit"), os.path.join(golden_dir, "barcode_cluster_map.tsv"))]
    # synthetic: fragments drawn around the peaks of the tokenizer's universe, 12 files x 3000 lines, 5 clusters
    rng = np.random.default_rng(3)
    peaks = [l.split()[:3] for l in open(os.path.join(golden_dir, "tokenizers", "peaks.bed")) if l.strip()]
    fd = tmp_path / "frags"
    fd.mkdir()
    barcodes = ["".join(rng.choice(list("ACGT"), 10)) for _ in range(30)]
    lines_map = []
    for fi in range(12):
        with gzip.open(fd / f"s{fi}.bed.gz", "wt") as f:
            for _ in range(3000):
                c, s, e = peaks[int(rng.integers(0, len(peaks)))]
                s2 = max(int(s) + int(rng.integers(-300, 300)), 0)
                f.write(f"{c}\t{s2}\t{s2 + int(rng.integers(1, 200))}\t{barcodes[int(rng.integers(0, 30))]}\t1\n")
        lines_map += [f"s{fi}+{b}\tk{(fi + i) % 5}" for i, b in enumerate(barcodes[:22])]
    mp = tmp_path / "map.tsv"
    mp.write_text("\n".join(lines_map) + "\n")
    cases.append((str(fd), str(mp)))
    for k, (files_dir, map_path) in enumerate(cases):
        m = BarcodeToClusterMap.from_file(map_path)
        fused = fragsplit_tokenize(files_dir, m, tk)
        out = tmp_path / f"split{k}"
        pseudobulk_fragment_files(files_dir, m, str(out))
        exp_text = oracle.fragsplit(files_dir, oracle.OracleBarcodeMap(map_path))
        assert sorted(fused) == sorted(exp_text)
        for label in m.cluster_labels():
            path = os.path.join(out, f"cluster_{label}.bed.gz")
            two_step = tokenize_fragment_file(path, tk)
            assert fused[label] == two_step, label
            assert list(fused[label]) == list(two_step)  # barcodes in first-seen order
            assert two_step == otk.tokenize_fragment_file(path), label


# ------------------------------------------------------------ LOLA universe helpers (gtars-lola/src/universe.rs:154-301)

def test_config5_at_the_per_file_size_against_the_compiled_restatement(tmp_path):
    """BASELINE config 5 at the config's per-file size -- 1e5 fragments and 500 barcodes per file, 40 files: more
    than one wave of the host pipeline on the test box, so the tokenizer calls overlap the parsing -- against the compiled C
    restatement of fragsplit + tokenize_fragment_file (oracle/fragsplit_oracle.c): ids, sum of ids and distinct barcodes of
    every cluster; and the first files id by id against the Python restatement."""
    from test_sharding_gloo import oracle_fragment_pipeline, same_cluster_results

    from gtars_amd import synth
    from gtars_amd.fragsplit import BarcodeToClusterMap, fragsplit_tokenize, fragsplit_tokenize_files, list_fragment_files
    from gtars_amd.tokenizers import Tokenizer

    u = synth.make_universe(100_000)
    ub, fd, mp, _ = synth.write_config5_inputs(str(tmp_path), u, 40, 100_000, 20)
    tok, m = Tokenizer.from_bed(ub), BarcodeToClusterMap.from_file(mp)
    om, otok = oracle.OracleBarcodeMap(mp), oracle.OracleTokenizer(ub)
    paths = list_fragment_files(fd)
    got = fragsplit_tokenize(fd, m, tok, as_arrays=True)
    exp = oracle.fragsplit_tokenize_compiled(paths, om, otok)
    assert sorted(got) == sorted(exp)
    for label, (names, offs, ids) in got.items():
        assert (int(offs[-1]), int(ids.astype(np.uint64).sum()), len(names)) == exp[label], label
    assert sum(v[0] for v in exp.values()) > 3_000_000
    few = paths[:2]
    assert same_cluster_results(fragsplit_tokenize_files(few, m, tok, as_arrays=True), oracle_fragment_pipeline(few, om, otok))


def test_fused_fragment_pipeline_errors_between_waves(tk, golden_dir, tmp_path, monkeypatch):
    """The fused pipeline with two host threads (waves of four files, the tokenizer calls on their helper thread): a malformed
    line in a LATER wave ends the call with the reference's error while an earlier wave is on the GPU -- no hang, nothing
    leaked into the next call, which returns the full result again."""
    import gzip

    from gtars_amd.fragsplit import BarcodeToClusterMap, fragsplit_tokenize

    monkeypatch.setenv("GTARS_HOST_THREADS", "2")
    tk = tk("peaks.bed")
    peaks = [l.split()[:3] for l in open(os.path.join(golden_dir, "tokenizers", "peaks.bed")) if l.strip()]
    rng = np.random.default_rng(11)
    fd = tmp_path / "frags"
    fd.mkdir()
    lines_map = []
    for fi in range(11):
        with gzip.open(fd / f"f{fi:02d}.bed.gz", "wt") as f:
            for _ in range(2000):
                c, s, e = peaks[int(rng.integers(0, len(peaks)))]
                f.write(f"{c}\t{int(s) + int(rng.integers(0, 50))}\t{int(e) + 5}\tBC{int(rng.integers(0, 9))}\t1\n")
        lines_map += [f"f{fi:02d}+BC{b}\tk{(fi + b) % 3}" for b in range(9)]
    mp = tmp_path / "map.tsv"
    mp.write_text("\n".join(lines_map) + "\n")
    m = BarcodeToClusterMap.from_file(str(mp))
    good = fragsplit_tokenize(str(fd), m, tk, as_arrays=True)
    assert sum(int(v[1][-1]) for v in good.values()) >= 11 * 2000
    with gzip.open(fd / "f09.bed.gz", "at") as f:  # (third wave) a routed line whose start is not a number
        f.write("chr1\tx12\t30\tBC1\t1\n")
    with pytest.raises(RuntimeError, match="Failed to parse start position of a routed fragment"):
        fragsplit_tokenize(str(fd), m, tk)
    with gzip.open(fd / "f05.bed.gz", "at") as f:  # (second wave) fewer than five fields: split.rs:84-90
        f.write("chr1\t12\t30\tBC1\n")
    with pytest.raises(RuntimeError, match="Failed to parse fragments file at line 2000"):
        fragsplit_tokenize(str(fd), m, tk)
    os.remove(fd / "f05.bed.gz")
    os.remove(fd / "f09.bed.gz")
    again = fragsplit_tokenize(str(fd), m, tk, as_arrays=True)
    assert 0 < sum(int(v[1][-1]) for v in again.values()) < sum(int(v[1][-1]) for v in good.values())



def test_fused_fragment_pipeline_gives_the_same_result_under_every_switch(tk, golden_dir, tmp_path, monkeypatch):
    """Round 5's host half of the fused pipeline has several forms, each behind a switch: files streamed to the device thread in
    batches that depend on thread timing (1 / 2 / many host threads), the library's own inflate decoder or zlib's
    (GTARS_ZLIB_INFLATE), the gzip members' CRC-32 on the device or on the host (GTARS_FRAG_HOST_CRC), text and results in blocks
    of the pinned pool, in ordinary memory (GTARS_NO_PINNED), in pinned blocks that are never cached (GTARS_PINNED_POOL_MB=0), or in a
    mix of both (GTARS_PINNED_MAX_MB=1: the pool hands out one block and refuses the rest),
    the host parser (GTARS_FRAG_HOST_PARSE).  Same per-cluster result -- barcodes in first-seen order, offsets, ids -- every time,
    and equal to the oracle's restatement of the two-step pipeline (split.rs:84-131 + fragments.rs:12-56)."""
    import gzip

    from gtars_amd import _lib
    from gtars_amd.fragsplit import BarcodeToClusterMap, fragsplit_tokenize, list_fragment_files
    from test_sharding_gloo import oracle_fragment_pipeline, same_cluster_results
    import oracle

    tok = tk("peaks.bed")
    ub = os.path.join(golden_dir, "tokenizers", "peaks.bed")
    peaks = [l.split()[:3] for l in open(ub) if l.strip()]
    rng = np.random.default_rng(23)
    fd = tmp_path / "frags"
    fd.mkdir()
    lines_map = []
    for fi in range(23):
        n = int(rng.integers(1, 4000)) if fi != 7 else 0  # (an empty file among them)
        text = "".join(f"{peaks[int(k)][0]}\t{int(peaks[int(k)][1]) + int(d)}\t{int(peaks[int(k)][2]) + 5}\tBC{int(b)}\t1\n"
                       for k, d, b in zip(rng.integers(0, len(peaks), n), rng.integers(0, 50, n), rng.integers(0, 12, n)))
        # levels 0 .. 9: stored, fixed and dynamic blocks; every third file as two gzip members
        blob = gzip.compress(text.encode(), compresslevel=fi % 10)
        if fi % 3 == 0 and n > 10:
            half = text[: len(text) // 2].rfind("\n") + 1
            blob = gzip.compress(text[:half].encode(), compresslevel=6) + gzip.compress(text[half:].encode(), compresslevel=1)
        (fd / f"f{fi:02d}.bed.gz").write_bytes(blob)
        lines_map += [f"f{fi:02d}+BC{b}\tk{(fi + b) % 4}" for b in range(9)]  # (BC9 .. BC11 are not mapped)
    mp = tmp_path / "map.tsv"
    mp.write_text("\n".join(lines_map) + "\n")
    m = BarcodeToClusterMap.from_file(str(mp))
    om, otok = oracle.OracleBarcodeMap(str(mp)), oracle.OracleTokenizer(ub)
    want = oracle_fragment_pipeline(list_fragment_files(str(fd)), om, otok)
    switches = [{}, {"GTARS_HOST_THREADS": "1"}, {"GTARS_HOST_THREADS": "2"}, {"GTARS_HOST_THREADS": "5"}, {"GTARS_ZLIB_INFLATE": "1"},
                {"GTARS_FRAG_HOST_CRC": "1"}, {"GTARS_NO_PINNED": "1"}, {"GTARS_PINNED_POOL_MB": "0"}, {"GTARS_PINNED_MAX_MB": "1"}, {"GTARS_FRAG_HOST_PARSE": "1"},
                {"GTARS_NO_PINNED": "1", "GTARS_ZLIB_INFLATE": "1", "GTARS_HOST_THREADS": "3"}]
    try:
        for sw in switches:
            for k, v in sw.items():
                monkeypatch.setenv(k, v)
            _lib.lib.gtars_debug_reload_env()
            for _ in range(2):  # (twice: the second call runs on whatever the first one left in the pools)
                got = fragsplit_tokenize(str(fd), m, tok, as_arrays=True)
                assert same_cluster_results(got, want), sw
            for k in sw:
                monkeypatch.delenv(k)
    finally:
        _lib.lib.gtars_debug_reload_env()


def test_device_fragment_parser_follows_the_reference_line_rules(tk, golden_dir, tmp_path, monkeypatch):
    """Round 5: the fused pipeline's text is split and parsed ON THE GPU (fragparse.hip).  Every line rule of
    gtars-fragsplit/src/split.rs:84-131 and gtars-tokenizers/src/utils/fragments.rs:12-82 on hand-made files, the device parser
    against the host parser (GTARS_FRAG_HOST_PARSE=1) and against the oracle's restatement of the two-step pipeline: CRLF line
    ends, a last line without a newline, runs of blanks and tabs between and in front of the fields, extra columns, '+' in front
    of a number, '#' chromosomes (routed, never tokenized), unknown chromosomes (tokenized to unk), barcodes that are not in the
    map (their lines are not even looked at: garbage numbers are fine there), an empty file, a plain-text (not gzip) file, the
    same barcode in two files under different clusters; then the failures, each with the reference's message whichever parser
    saw it first: fewer than five fields (also on unrouted lines, also an empty line), a routed line whose start / end is not
    a u32, and an unreadable file BEHIND a malformed one (the earlier file's error wins)."""
    import gzip

    import oracle
    from gtars_amd.fragsplit import BarcodeToClusterMap, fragsplit_tokenize, list_fragment_files
    from test_sharding_gloo import oracle_fragment_pipeline, same_cluster_results

    tok = tk("peaks.bed")
    ub = os.path.join(golden_dir, "tokenizers", "peaks.bed")
    peaks = [l.split()[:3] for l in open(ub) if l.strip()]
    rng = np.random.default_rng(5)
    fd = tmp_path / "frags"
    fd.mkdir()

    def frag(bc, chrom=None, plus=False):
        c, s, e = peaks[int(rng.integers(0, len(peaks)))]
        s, e = int(s) + int(rng.integers(0, 40)), int(e) + 3
        return (chrom or c, ("+" if plus else "") + str(s), str(e), bc, "2")

    a = ["\t".join(frag(f"A{k % 7}")) for k in range(300)]
    a[5] = "  " + "   ".join(frag("A1")) + "  \t extra col"  # blanks in front, runs of blanks, extra columns
    a[6] = "\t".join(frag("A2", plus=True))  # "+123" parses (str::parse::<u32>)
    a[7] = "\t".join(frag("A3", chrom="#chr1"))  # routed, not tokenized
    a[8] = "\t".join(frag("A4", chrom="chrNope"))  # unknown chromosome: one unk id
    a[9] = "chr1\tNaN\t-5\tZZ9\t1"  # a barcode that is not in the map: nothing else of the line matters
    a[10] = "chr1 100 200 A5 1 7 8 9"
    (fd / "a.bed").write_text("\r\n".join(a))  # CRLF, plain text, no newline behind the last line
    b = ["\t".join(frag(f"A{k % 5}")) for k in range(250)]  # the same barcodes as file a, other clusters
    with gzip.open(fd / "b.bed.gz", "wt") as f:
        f.write("\n".join(b) + "\n")
    with gzip.open(fd / "c.bed.gz", "wt") as f:
        f.write("")  # an empty file
    with gzip.open(fd / "d.tsv.gz", "wt") as f:  # two extensions: the stem is "d"
        f.write("\n".join("\t".join(frag("Q1")) for _ in range(40)) + "\n")
    mp = tmp_path / "map.tsv"
    mp.write_text("".join(f"a+A{k}\tc{k % 3}\n" for k in range(7)) + "".join(f"b+A{k}\tc{(k + 1) % 4}\n" for k in range(5)) + "d+Q1\tc9\n")
    m = BarcodeToClusterMap.from_file(str(mp))

    def both_parsers(check):
        check("device")
        monkeypatch.setenv("GTARS_FRAG_HOST_PARSE", "1")
        check("host")
        monkeypatch.delenv("GTARS_FRAG_HOST_PARSE")
        monkeypatch.setenv("GTARS_FRAG_DEVICE_WAVE_MB", "0")  # every wave "too large" for the device: the whole call again on the host
        check("device -> host")
        monkeypatch.delenv("GTARS_FRAG_DEVICE_WAVE_MB")

    om, otok = oracle.OracleBarcodeMap(str(mp)), oracle.OracleTokenizer(ub)
    expected = oracle_fragment_pipeline(list_fragment_files(str(fd)), om, otok)

    def good(which):
        got = fragsplit_tokenize(str(fd), m, tok, as_arrays=True)
        assert same_cluster_results(got, expected), which

    both_parsers(good)
    for threads in ("1", "3"):
        monkeypatch.setenv("GTARS_HOST_THREADS", threads)  # waves of one / three files
        both_parsers(good)
    monkeypatch.delenv("GTARS_HOST_THREADS")

    def expect_error(pattern):
        def check(which):
            with pytest.raises(RuntimeError, match=pattern):
                fragsplit_tokenize(str(fd), m, tok)
        both_parsers(check)

    keep = (fd / "b.bed.gz").read_bytes()
    for bad_line, pattern in (("chr1\t5\t9\tNOT_IN_MAP", "Failed to parse fragments file at line 250: chr1\t5\t9\tNOT_IN_MAP"),
                              ("", "Failed to parse fragments file at line 250: $"),
                              ("chr1\t1x\t9\tA1\t1", "Failed to parse start position of a routed fragment"),
                              ("chr1\t1\t4294967296\tA1\t1", "Failed to parse end position of a routed fragment"),
                              ("chr1\t\t\t5\t9\tA1", "Failed to parse fragments file at line 250")):
        with gzip.open(fd / "b.bed.gz", "wt") as f:
            f.write("\n".join(b) + "\n" + bad_line + "\nchr1\t1\t2\tA1\t1\n")
        expect_error(pattern)
    # an unreadable file behind the malformed one: files are visited in order, the malformed line is what the caller hears of
    (fd / "bb.bed.gz").write_bytes(b"\x1f\x8b this is not a gzip stream")
    expect_error("Failed to parse fragments file at line 250")
    (fd / "b.bed.gz").write_bytes(keep)
    expect_error("gzip|read error|Failed to")
    os.remove(fd / "bb.bed.gz")
    both_parsers(good)
    # gzip framing.  The device path inflates every member RAW on the host threads and checks its CRC-32 on the GPU: a file of
    # many members (bgzip-like, with an extra field, a name and a comment in the headers) gives the same result ...
    import io
    import struct

    body = ("\n".join(b) + "\n").encode()
    pieces = [body[i:i + 3000] for i in range(0, len(body), 3000)]
    multi = b""
    for k, piece in enumerate(pieces):
        member = gzip.compress(piece, 1 + k % 9)
        if k % 3 == 1:  # FEXTRA (bgzip's block-size field looks like this) + FNAME + FCOMMENT
            extra = b"BC\x02\x00\x34\x12"
            member = member[:3] + bytes([4 | 8 | 16]) + member[4:10] + struct.pack("<H", len(extra)) + extra + b"name.bed\x00" + b"a comment\x00" + member[10:]
        multi += member
    (fd / "b.bed.gz").write_bytes(multi)
    both_parsers(good)
    monkeypatch.setenv("GTARS_FRAG_HOST_CRC", "1")  # (the host's own CRC check instead of the device's)
    good("device parser, host CRC")
    monkeypatch.delenv("GTARS_FRAG_HOST_CRC")
    # ... a member whose CRC is not its trailer's is the reference's (zlib's, flate2's) data error, whoever computes the CRC
    bad_crc = bytearray(multi)
    first_len = len(gzip.compress(pieces[0], 1))
    bad_crc[first_len - 8] ^= 0x01  # the first member's CRC field
    (fd / "b.bed.gz").write_bytes(bytes(bad_crc))
    expect_error("gzip read error")
    # ... and so is a flipped bit in the LAST member's data (a valid deflate stream may survive it: then only the CRC tells)
    flipped = bytearray(keep)
    flipped[len(flipped) // 2] ^= 0x10
    (fd / "b.bed.gz").write_bytes(bytes(flipped))
    expect_error("gzip read error|Failed to parse")
    (fd / "b.bed.gz").write_bytes(keep)
    both_parsers(good)


def test_lola_universe_helpers_kats():
    from gtars_amd.lola import check_universe, redefine_user_sets

    # test_check_universe_full_coverage
    rep = check_universe([[("chr1", 100, 200), ("chr1", 2500, 2600)]], [("chr1", 0, 1000), ("chr1", 2000, 3000)])
    assert rep["totalRegions"] == [2] and rep["regionsInUniverse"] == [2] and abs(rep["coverage"][0] - 1.0) < 1e-10
    assert rep["manyToMany"] == [0] and rep["warnings"] == []
    # test_check_universe_low_coverage
    rep = check_universe([[("chr1", 50, 80), ("chr1", 500, 600), ("chr1", 700, 800)]], [("chr1", 0, 100)])
    assert rep["regionsInUniverse"] == [1] and abs(rep["coverage"][0] - 1.0 / 3.0) < 0.01 and rep["warnings"]
    assert "only 33.3% of regions overlap the universe" in rep["warnings"][0]
    # test_check_universe_many_to_many
    rep = check_universe([[("chr1", 120, 220)]], [("chr1", 100, 200), ("chr1", 150, 250)])
    assert rep["manyToMany"] == [1] and any("many-to-many" in w for w in rep["warnings"])
    # test_redefine_user_sets_basic / _dedup / _no_overlap
    uni = [("chr1", 100, 200), ("chr1", 300, 400), ("chr1", 500, 600)]
    assert redefine_user_sets([[("chr1", 150, 350)]], uni) == [[("chr1", 100, 200), ("chr1", 300, 400)]]
    assert redefine_user_sets([[("chr1", 120, 150), ("chr1", 200, 250)]], [("chr1", 100, 300)]) == [[("chr1", 100, 300)]]
    assert redefine_user_sets([[("chr1", 500, 600)]], [("chr1", 100, 200)]) == [[]]


def test_bench_py_two_ranks_on_one_gpu():
    """bench.py --gpus 2 under torch.distributed.run with the gloo backend (both ranks on this GPU): one JSON line from
    rank 0 with the whole-job value, the roofline object and the all-gatherv variant; small sizes."""
    import json
    import subprocess
    import sys

    from test_sharding_gloo import _free_port

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GTARS_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--queries", "200000", "--batches", "3", "--min-seconds", "0.01", "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak" and out["value"] > 0
    assert out["roofline"]["kernel"] == "k_tok_lds" and 0 < out["roofline"]["frac"] < 1
    assert out["with_allgather"]["value"] > 0 and out["timing"]["repetitions"] >= 11


def test_bench_py_launches_its_own_ranks():
    """Plain `python bench.py --gpus 2` (how the driver starts --gpus 1; no launcher, no WORLD_SIZE): the parent starts the two
    ranks itself.  gloo backend so that both ranks may share this box's one GPU; the chromosome-bucket objects run at 1/100 of
    the configs' sizes and their totals are checked against the oracle here."""
    import json
    import subprocess
    import sys

    import oracle
    from gtars_amd import synth

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["GTARS_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--queries", "200000",
           "--batches", "3", "--min-seconds", "0.01", "--no-cpu-baseline", "--scale-configs", "100"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size"] == 2 and out["backend"] == "gloo"
    assert sorted(x["rank"] for x in out["ranks"]) == [0, 1] and len({x["pid"] for x in out["ranks"]}) == 2
    assert out["verified"]
    # config 3 at 1/100: the sharded totals against the oracle on the same synthetic data
    ndb, nq, F = 500_000, 100_000, 1000
    db, q = synth.make_igd_db(ndb, F), synth.make_background_queries(nq)
    og = oracle.Igd()
    og.add_arrays(db["chrom"], db["start"], db["end"], np.arange(ndb, dtype=np.int32), db["file"])
    og.finalize()
    s3 = out["igd_config3_sharded"]
    assert s3["db_intervals"] == ndb and s3["queries"] == nq and sum(x["db_intervals"] for x in s3["per_rank"]) == ndb
    assert all(0 < x["db_intervals"] < ndb for x in s3["per_rank"])  # the database is cut, not replicated
    assert s3["pairwise"]["total_hits"] == int(og.count_set_overlaps(q["chrom"], q["start"], q["end"], 1, n_files=F).sum())
    assert s3["binary"]["total_hits"] == int(og.count_region_hits(q["chrom"], q["start"], q["end"], 1, n_files=F).sum())
    # config 4 at 1/100: support sum against the oracle
    s4 = out["lola_config4_sharded"]
    n_sets, per = 2000, 250
    db = synth.make_igd_db(n_sets * per, n_sets, seed=6)
    uni = synth.make_universe(10_000, seed=3)
    sel = np.sort(np.random.default_rng(9).choice(len(uni["chrom"]), 1000, replace=False))
    og = oracle.Igd()
    og.add_arrays(db["chrom"], db["start"], db["end"], np.arange(n_sets * per, dtype=np.int32), db["file"])
    og.finalize()
    sup = og.count_region_hits(uni["chrom"][sel], uni["start"][sel], uni["end"][sel], 1, n_files=n_sets)
    assert s4["support_sum"] == int(sup.sum()) and s4["user"] == 1000
    # config 5 at 1/100 per file: the file list is cut into two runs, the ranks' token ids add up to the single-process count
    s5 = out["fragsplit_config5_sharded"]
    runs = sorted(x["files"] for x in s5["per_rank"])
    assert runs[0][0] == 0 and runs[0][1] == runs[1][0] and runs[1][1] == s5["files"] and all(b > a for a, b in runs)
    assert s5["verified"] and s5["fragments"] == s5["files"] * s5["fragments_per_file"] and s5["value"] > 0
    # round 5: every N > 1 object says what each rank's GPU did (device time by HIP events) apart from the collective, and the
    # weak-scaled objects carry per-rank work of the single-GPU configs
    for o in (s3["pairwise"], s3["binary"], s4):
        pr = o["per_rank_ms"]
        assert len(pr) == 2 and all(x["device_ms"] > 0 and x["collective_ms"] >= 0 for x in pr), pr
    w3 = out["igd_config3_weak"]
    assert w3["scaling"] == "weak" and w3["db_intervals"] == 2 * ndb and w3["queries"] == 2 * nq
    db, q = synth.make_igd_db(ndb, F), synth.make_background_queries(nq)
    og = oracle.Igd()
    og.add_arrays(db["chrom"], db["start"], db["end"], np.arange(ndb, dtype=np.int32), db["file"])
    og.finalize()
    db1, q1 = synth.make_igd_db(ndb, F, seed=8), synth.make_background_queries(nq, seed=9)  # rank 1's share of the grown genome
    og1 = oracle.Igd()
    og1.add_arrays(db1["chrom"], db1["start"], db1["end"], np.arange(ndb, dtype=np.int32), db1["file"])
    og1.finalize()
    want = int(og.count_set_overlaps(q["chrom"], q["start"], q["end"], 1, n_files=F).sum()) + \
        int(og1.count_set_overlaps(q1["chrom"], q1["start"], q1["end"], 1, n_files=F).sum())
    assert w3["pairwise"]["total_hits"] == want and len(w3["pairwise"]["per_rank_ms"]) == 2
    w5 = out["fragsplit_config5_weak"]
    assert w5["scaling"] == "weak" and w5["files"] == 2 * max(2, 48 // 100) and w5["verified"] and w5["host_threads_per_rank"] >= 1
    # fewer devices than ranks under nccl: refused before anything runs
    env["GTARS_BENCH_BACKEND"] = "nccl"
    import torch

    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(torch.cuda.device_count() + 1)], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "visible devices" in r.stderr


def test_bench_py_with_eight_ranks_on_one_gpu():
    """`python bench.py --gpus 8` -- the command of the 8-GPU point of the scaling curve -- with the eight ranks sharing this box's
    one GPU under gloo, at 1/100 of the sharded configs: every rank reports, every sharded and weak-scaled object verifies
    itself (bench.py exits non-zero otherwise), the chromosome buckets leave no rank without database rows, and each rank's
    host-thread budget is its share of the box's."""
    import json
    import subprocess
    import sys

    import gtars_amd._lib as L

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env["GTARS_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--queries", "100000",
           "--batches", "2", "--min-seconds", "0.01", "--no-cpu-baseline", "--scale-configs", "100"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1800)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["world_size"] == 8 and sorted(x["rank"] for x in out["ranks"]) == list(range(8))
    assert len({x["pid"] for x in out["ranks"]}) == 8 and out["verified"]
    s3 = out["igd_config3_sharded"]
    assert len(s3["per_rank"]) == 8 and all(x["db_intervals"] > 0 for x in s3["per_rank"])
    assert sum(x["db_intervals"] for x in s3["per_rank"]) == s3["db_intervals"]
    for key in ("igd_config3_sharded", "igd_config3_weak"):
        for form in ("pairwise", "binary"):
            assert len(out[key][form]["per_rank_ms"]) == 8 and out[key][form]["total_hits"] > 0
    assert out["igd_config3_weak"]["queries"] == 8 * s3["queries"]
    assert len(out["lola_config4_sharded"]["per_rank_ms"]) == 8
    for key in ("fragsplit_config5_sharded", "fragsplit_config5_weak"):
        o = out[key]
        assert o["verified"] and sum(b - a for a, b in (x["files"] for x in o["per_rank"])) == o["files"]
    alone = L.lib.gtars_host_threads(64)
    assert out["fragsplit_config5_weak"]["host_threads_per_rank"] == max(1, min(64, L.lib.gtars_host_threads(0) // 8)), alone


# ------------------------------------------------------------ CLI text front ends (python -m gtars_amd overlaprs / igd)


def test_cli_overlaprs_text_matches_the_reference_rules(golden_dir, tmp_path):
    """gtars-cli/src/overlaprs/handlers.rs: same text, line for line, as the oracle restatement -- on the reference's
    tokenizer fixtures and on a synthetic universe / query pair (.gz, 30k x 50k lines, unknown chromosomes), both backends."""
    import gzip
    import io
    import subprocess
    import sys

    from gtars_amd import cli, synth

    peaks = os.path.join(golden_dir, "tokenizers", "peaks.bed")
    query = os.path.join(golden_dir, "to_tokenize.bed")
    for backend in ("bits", "ailist"):
        buf = io.StringIO()
        n = cli.run_overlaprs(peaks, query, backend, buf)
        exp = oracle.overlaprs_text(peaks, query, backend)
        assert buf.getvalue() == exp and n == exp.count("\n") and n > 0
    u = synth.make_universe(30_000, overlapping=True)
    q = synth.make_queries(u, 50_000, seed=4)
    up, qp = tmp_path / "u.bed.gz", tmp_path / "q.bed"
    with gzip.open(up, "wt") as fh:
        fh.write("".join(f"{synth.CHROM_NAMES[c]}\t{s}\t{e}\n" for c, s, e in zip(u["chrom"], u["start"], u["end"])))
    qp.write_text("".join(f"{synth.CHROM_NAMES[c] if c < synth.N_CHROM else 'chrUn_1'}\t{s}\t{e}\tname\n"
                          for c, s, e in zip(q["chrom"], q["start"], q["end"])))
    for backend in ("bits", "ailist"):
        buf = io.StringIO()
        cli.run_overlaprs(str(up), str(qp), backend, buf)
        assert buf.getvalue() == oracle.overlaprs_text(str(up), str(qp), backend)
    # the hit lines leave in chunks (a million per call of the C++ writer): chunk borders inside a query's hits and a binary sink
    chunk0 = cli.HIT_CHUNK
    try:
        cli.HIT_CHUNK = 7
        raw = io.BytesIO()
        buf = io.TextIOWrapper(raw, encoding="ascii", newline="")
        cli.run_overlaprs(str(up), str(qp), "bits", buf)
        buf.flush()
        assert raw.getvalue().decode() == oracle.overlaprs_text(str(up), str(qp), "bits")
        cli.HIT_CHUNK = 1
        buf = io.StringIO()
        cli.run_overlaprs(peaks, query, "ailist", buf)
        assert buf.getvalue() == oracle.overlaprs_text(peaks, query, "ailist")
    finally:
        cli.HIT_CHUNK = chunk0
    # through the command line
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "gtars_amd", "overlaprs", "-u", peaks, "-q", query], cwd=root, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and r.stdout == oracle.overlaprs_text(peaks, query, "bits")
    bad = tmp_path / "bad.bed"
    bad.write_text("chr1\t1 \t2\n")
    with pytest.raises(ValueError):
        cli.run_overlaprs(str(bad), query, "bits", io.StringIO())
    with pytest.raises(ValueError):
        cli.run_overlaprs(peaks, query, "nclist", io.StringIO())


def test_cli_igd_create_and_search(golden_dir, tmp_path):
    """gtars-cli/src/igd/handlers.rs: create from a directory / a .txt list, search prints the legacy TSV."""
    import io

    from gtars_amd import cli

    d = os.path.join(golden_dir, "igd_file_list_02")
    beds = sorted(os.path.join(d, n) for n in os.listdir(d) if n.endswith((".bed", ".gz")))
    assert cli.resolve_bed_paths(d) == beds
    lst = tmp_path / "files.txt"
    lst.write_text("\n".join(beds) + "\n\n")
    assert cli.resolve_bed_paths(str(lst)) == beds
    assert cli.resolve_bed_paths("-", io.StringIO("\n".join(beds))) == beds
    db_path = cli.run_igd_create(str(tmp_path), d, "mydb")
    assert db_path.endswith("mydb.igd") and os.path.exists(db_path)
    for qn in ("query1.bed", "query2.bed"):
        q = os.path.join(golden_dir, "igd_query_files", qn)
        buf = io.StringIO()
        total = cli.run_igd_search(db_path, q, buf)
        exp = oracle.igd_search_text(beds, q)
        assert buf.getvalue() == exp and f"Total: {total}\n" in exp
