"""Host layer checks that need no GPU: BED/BED.gz parsing + sort through the C ABI against the
oracle's restatement of RegionSet::try_from, .gtok bytes, error mapping."""
import os

import numpy as np
import pytest

import oracle

BEDS = [
    "tokenizers/peaks.bed", "tokenizers/peaks.bed.gz", "tokenizers/peaks.scored.bed", "to_tokenize.bed",
    "consensus/consensus1.bed", "igd_file_list_01/igd_bed_file_1.bed", "igd_file_list_02/igd_bed_file_2.bed",
    "igd_query_files/query1.bed", "igd_query_files/query2.bed", "regionset/dummy.bed", "regionset/dummy_b.bed",
    "regionset/dummy_headers.bed", "regionset/dummy.narrowPeak", "regionset/dummy.narrowPeak.bed.gz",
    "test_sorted_small.bed", "test_unsorted_small.bed", "test_unknown_chrom.bed",
    "lola_multi_db/collection1/regions/vistaEnhancers.bed", "fragments/region_scoring/fragments1.bed.gz",
]


@pytest.mark.parametrize("rel", BEDS)
def test_regionset_parse_matches_oracle(golden_dir, rel):
    from gtars_amd.models import RegionSet

    path = os.path.join(golden_dir, rel)
    try:
        exp = oracle.read_region_set(path)
    except oracle.RegionSetError:
        with pytest.raises(RuntimeError):
            RegionSet(path)
        return
    rs = RegionSet(path)
    got = [(r.chr, r.start, r.end, r.rest) for r in rs]
    assert got == exp
    assert len(rs) == len(exp)


def test_regionset_kats(golden_dir):
    from gtars_amd.models import Region, RegionSet

    # gtars-core/src/lib.rs:25-60: 25 regions, sorted (chr lexicographic, start)
    rs = RegionSet(os.path.join(golden_dir, "tokenizers", "peaks.bed"))
    assert len(rs) == 25
    keys = [(r.chr.encode(), r.start) for r in rs]
    assert keys == sorted(keys)
    # gtars-python/tests/test_regionset.py:17-29
    rs = RegionSet.from_regions([Region(chr="chr1", start=14, end=514, rest=None), Region(chr="chr19", start=19, end=810, rest=None)])
    assert isinstance(rs, RegionSet) and len(rs) == 2
    # from_regions does NOT sort (region_set.rs:212-220)
    rs = RegionSet.from_regions([Region("chr2", 5, 9, None), Region("chr1", 1, 2, "a\tb")])
    assert [str(r) for r in rs] == ["chr2\t5\t9", "chr1\t1\t2\ta\tb"]
    rs = RegionSet.from_vectors(["chr1", "chrX"], [1, 2], [5, 9])
    assert [r.chr for r in rs] == ["chr1", "chrX"] and rs.strands == ["*", "*"]
    with pytest.raises(ValueError):
        RegionSet.from_vectors(["chr1"], [1, 2], [5, 9])
    r = Region("chr1", 10, 25, None)
    assert len(r) == 15 and repr(r) == "Region -> chr1 10 25" and r == Region("chr1", 10, 25, "x")


def test_regionset_errors(tmp_path, golden_dir):
    from gtars_amd.models import RegionSet

    with pytest.raises(RuntimeError):
        RegionSet(str(tmp_path / "missing.bed"))
    empty = tmp_path / "empty.bed"
    empty.write_text("# only a comment\n")
    with pytest.raises(RuntimeError):  # EmptyRegionSet (region_set.rs:169-171)
        RegionSet(str(empty))
    bad = tmp_path / "bad.bed"
    bad.write_text("chr1\t10\t20\nchr1\tx\t30\n")
    with pytest.raises(RuntimeError):
        RegionSet(str(bad))
    # header line without '#': skipped because column 2 is not numeric (region_set.rs:120-135)
    hdr = tmp_path / "hdr.bed"
    hdr.write_text("chrom\tstart\tend\nchr2\t5\t9\nchr1\t7\t8\r\n")
    rs = RegionSet(str(hdr))
    assert [(r.chr, r.start, r.end) for r in rs] == [("chr1", 7, 8), ("chr2", 5, 9)]
    assert rs.header == "chrom\tstart\tend"


def test_gtok_roundtrip_and_golden_bytes(golden_dir, tmp_path):
    from gtars_amd import utils

    assert utils.read_tokens_from_gtok(os.path.join(golden_dir, "out", "peaks.gtok")) == list(range(25))
    assert utils.read_tokens_from_gtok(os.path.join(golden_dir, "out", "tokens.gtok")) == [42, 101, 999]
    p = str(tmp_path / "sub" / "dir" / "t.gtok")
    utils.write_tokens_to_gtok(p, list(range(25)))
    assert open(p, "rb").read() == open(os.path.join(golden_dir, "out", "peaks.gtok"), "rb").read()
    utils.write_tokens_to_gtok(p, [1, 70000, 3])
    assert open(p, "rb").read()[:5] == b"GTOK\x02"
    assert utils.read_tokens_from_gtok(p) == [1, 70000, 3]
    assert utils.read_tokens_from_gtok_as_strings(p) == ["1", "70000", "3"]
    # same bytes as the oracle's writer
    q = str(tmp_path / "o.gtok")
    oracle.write_tokens_to_gtok(q, [1, 70000, 3])
    assert open(p, "rb").read() == open(q, "rb").read()
    bad = tmp_path / "bad.gtok"
    bad.write_bytes(b"NOPE\x01\x00\x00")
    with pytest.raises(ValueError):
        utils.read_tokens_from_gtok(str(bad))


def test_lola_statistics_tail_kats():
    from gtars_amd import lola

    # enrichment.rs:498-515: CMLE vs R fisher.test(matrix(c(10,30,20,40), nrow=2))$estimate = 0.6693434
    assert abs(lola.odds_ratio(10, 20, 30, 40) - 0.6693434) < 1e-3
    # enrichment.rs:517-537 boundaries
    assert lola.odds_ratio(10, 0, 5, 100) == float("inf")
    assert lola.odds_ratio(0, 5, 10, 100) == 0.0
    # enrichment.rs:538-575: p-value inequalities only (statrs absent: parity unpinned, scipy used)
    assert lola.fisher_pvalue(50, 10, 10, 1000, True) < 1e-10
    assert lola.fisher_pvalue(1, 100, 100, 10, False) < 0.01
    assert lola.fisher_pvalue(0, 5, 5, 100, True) == 1.0
    rows = [{"userSet": 0, "dbSet": i, "pValueLog": p, "oddsRatio": o, "support": s} for i, (p, o, s) in
            enumerate([(5.0, 2.0, 10), (5.0, float("nan"), 10), (1.0, 3.0, 20)])]
    lola._rank_results(rows)
    assert [r["rnkPV"] for r in rows] == [1, 1, 3]        # ties.method = "min"
    assert [r["rnkOR"] for r in rows] == [2, 3, 1]        # NaN ranks last
    assert [r["rnkSup"] for r in rows] == [2, 2, 1]
    lola._apply_fdr(rows)
    assert all(0.0 <= r["qValue"] <= 1.0 for r in rows)


def _py_parse_fragments(text):
    rows = []
    for line in text.split("\n")[:-1] if text.endswith("\n") else text.split("\n"):
        if line.endswith("\r"):
            line = line[:-1]
        if line.startswith("#"):
            continue
        p = line.split()
        rows.append((p[0], int(p[1]), int(p[2]), p[3]))
    return rows


@pytest.mark.parametrize("threads", ["1", "3", "8"])
def test_fragment_reader_matches_a_plain_python_parse(tmp_path, monkeypatch, golden_dir, threads):
    """gtars_fragments_read: SoA columns + first-seen-order dictionaries, whatever the thread count."""
    import gzip

    from gtars_amd import utils

    monkeypatch.setenv("GTARS_HOST_THREADS", threads)
    rng = np.random.default_rng(5)
    n = 120_000   # > 1 MB of text, so that several chunks are cut
    chroms = ["chr1", "chr10", "chr2", "chrX", "chrUn_x"]
    c = np.sort(rng.integers(0, len(chroms), n))
    s = rng.integers(0, 2**32 - 700, n, dtype=np.uint64)
    e = s + rng.integers(1, 600, n).astype(np.uint64)
    b = rng.integers(0, 300, n)
    lines = [f"{chroms[a]}\t{x}\t{y}\tBC{z:04d}-1\t{1 + z % 3}" for a, x, y, z in zip(c, s, e, b)]
    lines[1000] = "# a comment in the middle"
    lines[5] = "chr1   77 \t +88  BCspace   1   extra fields are fine"
    text = "\n".join(lines) + "\r\n"
    plain = tmp_path / "f.tsv"
    plain.write_text(text)
    gz = tmp_path / "f.tsv.gz"
    with gzip.open(gz, "wt") as fh:
        fh.write(text)
    exp = _py_parse_fragments(text)
    for path in (plain, gz):
        d = utils.read_fragments(str(path))
        got = list(zip([d["chrom_names"][i] for i in d["chrom"]], d["start"].tolist(), d["end"].tolist(),
                       [d["barcode_names"][i] for i in d["barcode"]]))
        assert got == exp
        # dictionaries are in first-seen order
        seen = []
        for row in exp:
            if row[3] not in seen:
                seen.append(row[3])
        assert d["barcode_names"] == seen
    # the golden fragment fixture of the reference's scoring tests
    d = utils.read_fragments(os.path.join(str(golden_dir), "fragments/region_scoring/fragments1.bed.gz"))
    import gzip as _gz
    with _gz.open(os.path.join(str(golden_dir), "fragments/region_scoring/fragments1.bed.gz"), "rt") as fh:
        exp = _py_parse_fragments(fh.read())
    assert list(zip(d["start"].tolist(), d["end"].tolist())) == [(r[1], r[2]) for r in exp]


def test_fragment_reader_errors_carry_the_reference_line_numbers(tmp_path, monkeypatch):
    from gtars_amd import utils

    monkeypatch.setenv("GTARS_HOST_THREADS", "4")
    good = "chr1\t10\t20\tAAAC-1\t1\n"
    n = 60_000  # several chunks
    for bad, msg in (("chr1\t10\t20\tAAAC-1\n", "Invalid fragment file detected at line: 41234"),
                     ("chr1\tx10\t20\tAAAC-1\t1\n", "Failed to parse start position at line 41234"),
                     ("chr1\t10\t4294967296\tAAAC-1\t1\n", "Failed to parse end position at line 41234")):
        p = tmp_path / "bad.tsv"
        p.write_text(good * 41234 + bad + good * (n - 41235))
        with pytest.raises(RuntimeError) as ei:
            utils.read_fragments(str(p))
        assert msg in str(ei.value)
    with pytest.raises(RuntimeError):
        utils.read_fragments(str(tmp_path / "missing.tsv"))
    empty = tmp_path / "empty.tsv"
    empty.write_text("")
    assert len(utils.read_fragments(str(empty))["chrom"]) == 0


@pytest.mark.parametrize("threads", ["1", "5"])
def test_regionset_multichunk_bed_matches_oracle(tmp_path, monkeypatch, threads):
    """RegionSet::try_from on a file big enough to be cut into several parser chunks: same regions, same
    stable (chr bytes, start) order, same header text as the oracle -- whatever the thread count."""
    from gtars_amd.models import RegionSet

    monkeypatch.setenv("GTARS_HOST_THREADS", threads)
    rng = np.random.default_rng(9)
    n = 90_000
    chroms = ["chr2", "chr10", "chr1", "chrX", "chr1_alt", "chrM"]
    c = rng.integers(0, len(chroms), n)
    s = rng.integers(0, 5_000, n)          # many start ties: the sort must be stable
    e = s + rng.integers(0, 300, n)
    lines = []
    for i in range(n):
        kind = i % 4
        rest = "" if kind == 0 else f"\tname{i}" if kind == 1 else f"\tname{i}\t{i % 7}\t+" if kind == 2 else "\t"
        lines.append(f"{chroms[c[i]]}\t{s[i]}\t{e[i]}{rest}")
    lines.insert(0, "chrom\tstart\tend\tname")   # column header without '#': first line only
    lines.insert(40_000, "# a comment in the middle")
    lines.insert(70_000, "track name=x")
    text = "\n".join(lines) + "\r\n"
    p = tmp_path / "big.bed"
    p.write_text(text)
    exp = oracle.read_region_set(str(p))
    rs = RegionSet(str(p))
    got = [(r.chr, r.start, r.end, r.rest) for r in rs]
    assert got == exp
    # errors keep the reference's message with the offending line
    bad = tmp_path / "bad.bed"
    bad.write_text("\n".join(lines[:60_000] + ["chr1\t12\tx13\tboom"] + lines[60_000:]) + "\n")
    with pytest.raises(RuntimeError) as ei:
        RegionSet(str(bad))
    assert "Error in parsing end position" in str(ei.value) and "boom" in str(ei.value)


# ------------------------------------------------------------ gtars-fragsplit (host only: no GPU involved)


def _cluster_text(out_dir, label):
    import gzip

    with gzip.open(os.path.join(out_dir, f"cluster_{label}.bed.gz"), "rt") as f:
        return f.read()


def test_fragsplit_reference_fixtures(golden_dir, tmp_path):
    """pseudobulk_fragment_files on the reference's own fixtures (split.rs tests: fragments/fragsplit + barcode_cluster_map.tsv)
    against the oracle restatement; map.rs KATs: 3 cluster labels, a QC-dropped barcode maps to nothing."""
    from gtars_amd.fragsplit import BarcodeToClusterMap, pseudobulk_fragment_files

    mp = os.path.join(golden_dir, "barcode_cluster_map.tsv")
    fd = os.path.join(golden_dir, "fragments", "fragsplit")
    m, om = BarcodeToClusterMap.from_file(mp), oracle.OracleBarcodeMap(mp)
    assert m.n_clusters() == 3 == len(om.cluster_labels) and m.get_cluster_labels() == om.cluster_labels
    assert len(m) == len(om.map)
    assert m.get_cluster_from_barcode("AAACGCAAGCAAAGGATCGGCT") is None
    for k, v in om.map.items():
        assert m.get_cluster_from_barcode(k) == v
    stats = pseudobulk_fragment_files(fd, m, str(tmp_path / "out" / "nested"))
    exp = oracle.fragsplit(fd, om)
    assert stats["written"] == sum(len(v) for v in exp.values()) and stats["reads"] == 30
    for label in m.cluster_labels():
        assert _cluster_text(tmp_path / "out" / "nested", label) == "".join(exp[label])


def test_fragsplit_randomized_and_errors(tmp_path):
    """many files (plain and .gz, CRLF, no trailing newline, extra columns, multi-dot names), duplicate map keys (the later
    line wins), clusters without reads (their file exists and is empty), and the reference's error messages."""
    import gzip

    from gtars_amd.fragsplit import BarcodeToClusterMap, pseudobulk_fragment_files

    rng = np.random.default_rng(11)
    fd = tmp_path / "frags"
    fd.mkdir()
    barcodes = ["".join(rng.choice(list("ACGT"), 12)) for _ in range(40)]
    map_lines = []
    for fi in range(23):
        name = f"sample{fi}.v2.bed" + (".gz" if fi % 2 else "")
        rows = []
        for _ in range(int(rng.integers(0, 400))):
            b = barcodes[int(rng.integers(0, len(barcodes)))]
            s = int(rng.integers(0, 1_000_000))
            sep = "\t" if rng.random() < 0.8 else "  "
            extra = f"{sep}extra" if rng.random() < 0.1 else ""
            rows.append(sep.join([f"chr{int(rng.integers(1, 5))}", str(s), str(s + 50), b, str(int(rng.integers(1, 4)))]) + extra)
        text = ("\r\n" if fi % 5 == 0 else "\n").join(rows) + ("" if fi % 3 == 0 else "\n")
        if name.endswith(".gz"):
            with gzip.open(fd / name, "wt", newline="") as f:
                f.write(text)
        else:
            with open(fd / name, "w", newline="") as f:
                f.write(text)
        for b in barcodes[: 25 + fi % 7]:
            map_lines.append(f"sample{fi}+{b}\t{'c' + str(hash((fi, b)) % 6)}")
    map_lines.append(f"sample0+{barcodes[0]} override")   # later line wins, new label
    map_lines.append("nobody+AAAA lonely")                  # a cluster that receives nothing
    mp = tmp_path / "map.tsv"
    mp.write_text("\n".join(map_lines))                     # no trailing newline
    m, om = BarcodeToClusterMap.from_file(str(mp)), oracle.OracleBarcodeMap(str(mp))
    assert m.get_cluster_from_barcode(f"sample0+{barcodes[0]}") == "override"
    assert m.get_cluster_labels() == om.cluster_labels and "lonely" in om.cluster_labels
    out = tmp_path / "out"
    stats = pseudobulk_fragment_files(str(fd), m, str(out))
    exp = oracle.fragsplit(str(fd), om)
    assert stats["written"] == sum(len(v) for v in exp.values()) > 1000
    for label in m.cluster_labels():
        assert _cluster_text(out, label) == "".join(exp[label]), label
    assert _cluster_text(out, "lonely") == ""
    # errors: a short line (0-based index, the line itself), a bad map line, a missing directory
    (fd / "zzz_bad.bed").write_text("chr1\t1\t2\tAAAA\t1\nchr1\t5\t6\tAAAA\n")
    with pytest.raises(RuntimeError, match=r"Failed to parse fragments file at line 1: chr1\t5\t6\tAAAA"):
        pseudobulk_fragment_files(str(fd), m, str(tmp_path / "out2"))
    with pytest.raises(ValueError, match="line 1"):
        oracle.fragsplit(str(fd), om)
    bad = tmp_path / "bad.tsv"
    bad.write_text("a+b 1\nonlyone\n")
    with pytest.raises(RuntimeError, match="Invalid line format"):
        BarcodeToClusterMap.from_file(str(bad))
    with pytest.raises(RuntimeError, match="error reading the specifed fragment file directory"):
        pseudobulk_fragment_files(str(tmp_path / "nope"), m, str(tmp_path / "out3"))


def test_remove_all_extensions_matches_reference_rule():
    # gtars-core/src/utils.rs:372-387 through the product's lookup key: "<stem>+<barcode>"
    assert oracle.remove_all_extensions("/x/y/fragments1.bed.gz") == "fragments1"
    assert oracle.remove_all_extensions("a.b.c.d") == "a"
    assert oracle.remove_all_extensions(".hidden") == ".hidden"
    assert oracle.remove_all_extensions("plain") == "plain"


@pytest.mark.parametrize("threads", [1, 5])
def test_cli_bed3_text_mode_reader(tmp_path, monkeypatch, threads):
    """gtars_bed3_lines_read (the overlaprs front end's rules, gtars-cli/src/overlaprs/handlers.rs:64-92): every line counts,
    TAB-only split, str::parse::<u32> -- against a plain Python statement of those rules, multi-chunk, with the errors."""
    from gtars_amd import cli

    monkeypatch.setenv("GTARS_HOST_THREADS", str(threads))
    rng = np.random.default_rng(5)
    n = 120_000
    chroms = [f"chr{rng.integers(1, 30)}" if i % 997 else "#odd name" for i in range(n)]
    starts, ends = rng.integers(0, 2**32, n), rng.integers(0, 2**32, n)
    lines = [f"{c}\t{'+' if i % 13 == 0 else ''}{s}\t{e}" + ("\trest\tmore" if i % 3 == 0 else "") + ("\r" if i % 7 == 0 else "")
             for i, (c, s, e) in enumerate(zip(chroms, starts, ends))]
    p = tmp_path / "q.bed"
    p.write_text("\n".join(lines))  # no trailing newline: the last line still counts
    names, cid, s, e = cli.read_bed3_lines(str(p))
    assert [names[k] for k in cid] == chroms and s.tolist() == starts.tolist() and e.tolist() == ends.tolist()
    first_seen = list(dict.fromkeys(chroms))
    assert names == first_seen
    for bad, msg in (("chr1\t5", "Missing end field"), ("chr1", "Missing start field"), ("", "Missing start field"),
                     ("chr1\t-1\t5", "invalid digit"), ("chr1\t1\t5 ", "invalid digit"), ("chr1\t1\t4294967296", "invalid digit"),
                     ("chr1 1 5", "Missing start field")):
        q = tmp_path / "bad.bed"
        q.write_text("\n".join(lines[:70_000] + [bad] + lines[70_000:]))
        with pytest.raises(ValueError, match=f"bad.bed:70001: {msg}"):
            cli.read_bed3_lines(str(q))


def test_dense_region_ids_and_hit_lines(tmp_path):
    """gtars_regionset_dense_ids = generate_region_to_id_map (gtars-core/src/utils.rs:202-214: first-seen ids over the whole
    Region incl. rest); gtars_format_hit_lines = the overlaprs output lines."""
    import ctypes as C

    from gtars_amd import _lib
    from gtars_amd.models import RegionSet

    p = tmp_path / "c.bed"
    p.write_text("chr1\t10\t20\ta\nchr1\t10\t20\tb\nchr1\t10\t20\ta\nchr2\t5\t9\nchr2\t5\t9\nchr1\t1\t2\n")
    rs = RegionSet(str(p))
    regs = [(r.chr, r.start, r.end, r.rest) for r in rs]
    exp, seen = [], {}
    for r in regs:
        exp.append(seen.setdefault(r, len(seen)))
    h, n_ids = C.c_void_p(), C.c_uint32()
    _lib.check(_lib.lib.gtars_regionset_dense_ids(rs._h, C.byref(h), C.byref(n_ids)))
    assert _lib.take_u32(h, len(rs)).tolist() == exp and n_ids.value == len(seen) == 4
    names = [b"chrA", b"chr10"]
    arr = (C.c_char_p * 2)(*names)
    hc, hs, he = (np.asarray(x, dtype=np.uint32) for x in ([1, 0, 1], [0, 4294967295, 12], [7, 5, 4000000000]))
    text, ln = C.c_void_p(), C.c_uint64()
    _lib.check(_lib.lib.gtars_format_hit_lines(C.cast(arr, C.c_void_p), _lib.ptr(hc), _lib.ptr(hs), _lib.ptr(he), 3, C.byref(text), C.byref(ln)))
    try:
        assert C.string_at(text, ln.value) == b"chr10\t0\t7\nchrA\t4294967295\t5\nchr10\t12\t4000000000\n"
    finally:
        _lib.lib.gtars_free(text)


def test_scoring_output_writers(tmp_path):
    """write_sparse_counts_to_mtx (gtars-scoring/src/matrix_market.rs:26-92) and CountMatrix::write_to_file (counts.rs:89-105):
    the exact text the reference writes -- barcodes in byte order, triplets sorted by (row, col), 1-based, `peak_<i>` features;
    rows of a dense matrix joined by commas."""
    import gzip

    from gtars_amd import scoring

    prefix = str(tmp_path / "m")
    scoring.write_sparse_counts_to_mtx({"TTG": {2: 5}, "AAC": {3: 1, 0: 2}, "Abc": {}}, 4, prefix)
    assert gzip.open(prefix + "_matrix.mtx.gz", "rt").read() == ("%%MatrixMarket matrix coordinate integer general\n3 4 3\n"
                                                                  "1 1 2\n1 4 1\n3 3 5\n")
    assert gzip.open(prefix + "_barcodes.tsv.gz", "rt").read() == "AAC\nAbc\nTTG\n"
    assert gzip.open(prefix + "_features.tsv.gz", "rt").read() == "peak_0\npeak_1\npeak_2\npeak_3\n"
    scoring.write_count_matrix(np.array([[2, 2, 1, 3], [4, 1, 3, 1]], dtype=np.uint32), str(tmp_path / "c.csv.gz"))
    assert gzip.open(tmp_path / "c.csv.gz", "rt").read() == "2,2,1,3\n4,1,3,1\n"


def test_host_thread_budget_is_shared_between_the_ranks_of_a_node(monkeypatch):
    """every rank of a launcher gets its SHARE of the node's host threads (LOCAL_WORLD_SIZE, torch.distributed.run): the fragment
    pipeline is host-bound, eight ranks that each start every thread the node has only fight over the cores; GTARS_HOST_THREADS
    (a launcher's per-rank setting) overrides; `cap` bounds everything but the override"""
    import gtars_amd._lib as L

    monkeypatch.delenv("GTARS_HOST_THREADS", raising=False)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    alone = L.lib.gtars_host_threads(0)
    assert alone >= 1
    assert L.lib.gtars_host_threads(3) == min(alone, 3)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert L.lib.gtars_host_threads(0) == max(1, alone // 8)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
    assert L.lib.gtars_host_threads(0) == alone
    monkeypatch.setenv("GTARS_HOST_THREADS", "5")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert L.lib.gtars_host_threads(2) == 5


def test_gzip_reader_follows_the_multi_member_decoder(tmp_path):
    """Round 5: the host layer inflates with zlib's inflate() on the whole compressed file (the gz layer of zlib cost half as
    much again as the inflate).  What a reader of gtars-core/src/utils.rs:115-126 (flate2's MultiGzDecoder behind the "gz"
    extension) and zlib's gzread both do: concatenated members are decoded one after the other, a large file (several output
    growth steps: its ISIZE lies about the length when members are concatenated), bytes behind the last member that do not
    start another one are an error (MultiGzDecoder: "invalid gzip header"; round 5 ignored them like gzread), a ".gz" without the gzip magic is read as it is, an empty file is empty; a file that ends
    inside a member, or whose CRC is wrong, is an error."""
    import gzip

    from gtars_amd import utils

    rng = np.random.default_rng(3)
    lines = [f"chr{1 + i // 4000}\t{1000 + 37 * i}\t{1200 + 37 * i}\tBC{int(rng.integers(0, 300)):03d}\t{1 + i % 3}\n" for i in range(20_000)]
    text = "".join(lines)

    def check(path, n_lines):
        d = utils.read_fragments(str(path))
        assert len(d["start"]) == n_lines
        if n_lines:
            assert int(d["start"][0]) == 1000 and int(d["start"][n_lines - 1]) == 1000 + 37 * ((n_lines - 1) % 20_000)

    one = gzip.compress(text.encode())
    (tmp_path / "one.bed.gz").write_bytes(one)
    check(tmp_path / "one.bed.gz", 20_000)
    # three members (the last one's ISIZE says a third of the total), then the same with garbage behind the last member
    (tmp_path / "three.bed.gz").write_bytes(one + gzip.compress(text.encode(), 1) + one)
    check(tmp_path / "three.bed.gz", 60_000)
    (tmp_path / "tail.bed.gz").write_bytes(one + b"\x00\x00garbage that is no gzip header")
    with pytest.raises((RuntimeError, ValueError, OSError), match="gzip read error: invalid gzip header"):
        utils.read_fragments(str(tmp_path / "tail.bed.gz"))  # (MultiGzDecoder fails on it; gzread would ignore it)
    (tmp_path / "plain.bed.gz").write_text(text)  # no magic: passed through
    check(tmp_path / "plain.bed.gz", 20_000)
    (tmp_path / "empty.bed.gz").write_bytes(b"")
    check(tmp_path / "empty.bed.gz", 0)
    (tmp_path / "cut.bed.gz").write_bytes(one[: len(one) // 2])
    with pytest.raises((RuntimeError, ValueError, OSError), match="gzip read error"):
        utils.read_fragments(str(tmp_path / "cut.bed.gz"))
    bad = bytearray(one)
    bad[-6] ^= 0x55  # the CRC in the trailer
    (tmp_path / "crc.bed.gz").write_bytes(bytes(bad))
    with pytest.raises((RuntimeError, ValueError, OSError), match="gzip read error"):
        utils.read_fragments(str(tmp_path / "crc.bed.gz"))


def _read_file(path):
    from gtars_amd import utils

    return utils.read_file(path)


@pytest.mark.parametrize("decoder", ["fast", "zlib"])
def test_whole_buffer_inflate_decodes_what_zlib_writes(tmp_path, monkeypatch, decoder):
    """Round 5: the deflate streams of a ".gz" are decoded by the library's own whole-buffer decoder (csrc/inflate_fast.h; zlib's
    behind GTARS_ZLIB_INFLATE) -- gtars_read_file, the get_dynamic_reader of gtars-core/src/utils.rs:115-126 as one call, must
    return the bytes that went in for every block type zlib can write: stored (level 0), fixed codes (Z_FIXED), dynamic codes
    at several levels / strategies / memory levels (short and long codes, long matches, distance 1 runs, incompressible bytes),
    an optional-field header (FEXTRA as bgzip writes it, FNAME, FCOMMENT, FHCRC), concatenated and empty members; damaged
    files are errors with zlib's messages (the decoder refuses, zlib diagnoses).  tests/soak/fuzz_inflate.cpp is the long form."""
    import zlib

    from gtars_amd import _lib

    if decoder == "zlib":
        monkeypatch.setenv("GTARS_ZLIB_INFLATE", "1")
    _lib.lib.gtars_debug_reload_env()
    try:
        rng = np.random.default_rng(11)

        def gz(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem=8):
            c = zlib.compressobj(level, zlib.DEFLATED, 31, mem, strategy)
            return c.compress(data) + c.flush()

        frag = "".join(f"chr{1 + int(c)}\t{int(s)}\t{int(s) + 300}\tBC{int(b):05d}\t1\n"
                       for c, s, b in zip(rng.integers(0, 22, 30_000), rng.integers(0, 10**8, 30_000), rng.integers(0, 500, 30_000))).encode()
        samples = [b"", b"a", b"a" * 100_000, b"ab" * 70_000, b"abc" * 50_000 + b"xyz", rng.integers(0, 256, 70_000, dtype=np.uint8).tobytes(),
                   rng.integers(0, 4, 200_000, dtype=np.uint8).tobytes(), frag, (b"0123456" * 9 + b"\n") * 40_000,
                   rng.integers(0, 256, 1_200_000, dtype=np.uint8).tobytes()]
        p = tmp_path / "x.bed.gz"
        for k, data in enumerate(samples):
            for level in (0, 1, 6, 9):
                for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                    for mem in (1, 8):
                        p.write_bytes(gz(data, level, strategy, mem))
                        assert _read_file(p) == data, (k, level, strategy, mem)
        # optional header fields, three members (one of them empty)
        a, b = b"hello\n" * 1000, rng.integers(0, 256, 5000, dtype=np.uint8).tobytes()
        raw = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = raw.compress(a) + raw.flush()
        hdr = bytes([0x1F, 0x8B, 8, 4 | 8 | 16 | 2, 0, 0, 0, 0, 0, 3]) + bytes([6, 0]) + b"BC\x02\x00\x12\x34" + b"name\x00" + b"comment\x00"
        tail = body + zlib.crc32(a).to_bytes(4, "little") + len(a).to_bytes(4, "little")
        p.write_bytes(hdr + (zlib.crc32(hdr) & 0xFFFF).to_bytes(2, "little") + tail + gz(b) + gz(b""))
        assert _read_file(p) == a + b
        p.write_bytes(hdr + ((zlib.crc32(hdr) ^ 1) & 0xFFFF).to_bytes(2, "little") + tail)
        with pytest.raises((RuntimeError, ValueError, OSError), match="gzip read error: header crc mismatch"):
            _read_file(p)
        p.write_bytes(gz(a) + b"garbage")  # (gzread: ignored; flate2's MultiGzDecoder, the reference's reader: an error)
        with pytest.raises((RuntimeError, ValueError, OSError), match="gzip read error: invalid gzip header"):
            _read_file(p)
        p.write_bytes(gz(a) + b"\x1f")  # one stray byte
        with pytest.raises((RuntimeError, ValueError, OSError), match="gzip read error: invalid gzip header"):
            _read_file(p)
        blob = bytearray(gz(frag))
        blob[-5] ^= 1
        p.write_bytes(bytes(blob))
        with pytest.raises((RuntimeError, ValueError, OSError), match="gzip read error: incorrect data check"):
            _read_file(p)
        p.write_bytes(gz(frag)[:-20])
        with pytest.raises((RuntimeError, ValueError, OSError), match="gzip read error: unexpected end of file"):
            _read_file(p)
        blob = gz(frag)
        rejected = 0
        for pos in range(20, len(blob) - 8, 4001):
            bad = bytearray(blob)
            bad[pos] ^= 0x55
            p.write_bytes(bytes(bad))
            try:
                got = _read_file(p)
                assert got == frag  # (a flipped bit that changes nothing does not exist in a deflate stream, but be exact)
            except (RuntimeError, ValueError, OSError):
                rejected += 1
        assert rejected >= (len(blob) - 28) // 4001
    finally:
        monkeypatch.delenv("GTARS_ZLIB_INFLATE", raising=False)
        _lib.lib.gtars_debug_reload_env()


def test_inflate_decoder_under_the_sanitizers(tmp_path):
    """tests/soak/fuzz_inflate.cpp, a short run, built with AddressSanitizer + UBSan (the GPU pool has no sanitizers: the decoder is
    host code, so it is checked here): random data kinds x zlib level / strategy / window / memory level / flush points decode
    bit-exact, and damaged streams (flipped bits, cut tails) never take the decoder out of its buffers."""
    import shutil
    import subprocess

    cxx = shutil.which("g++") or shutil.which("c++")
    if not cxx:
        pytest.skip("no host C++ compiler")
    exe = tmp_path / "fuzz_inflate"
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "soak", "fuzz_inflate.cpp")
    build = subprocess.run([cxx, "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", str(exe), src, "-lz"],
                           capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("the host compiler has no sanitizer runtime")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([str(exe), "250", "77"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, (run.stdout + run.stderr)[-2000:]
    assert "decoded bit-exact" in run.stdout


def test_bench_traffic_is_each_counters_mean_over_its_own_pass(tmp_path):
    """bench.py's roofline.traffic: FETCH_SIZE and WRITE_SIZE come from two SEPARATE rocprofv3 passes that may launch the kernel
    a different number of times; each counter's bytes per launch is its own sum over its own dispatch count (round 5 divided
    both sums by the FETCH_SIZE pass's count, which scaled the write bytes by 550 / 450)."""
    import bench

    head = '"Correlation_Id","Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"\n'
    kname = "void gtars::k_tok_lds<1024, 4, 1, 1, false, true, false, 5>(gtars::AccelView, unsigned int const*)"
    other = "void gtars::k_scan_blocks(unsigned int const*)"

    def write(path, counter, n_tok, kb, n_other):
        with open(path, "w") as f:
            f.write(head)
            for i in range(n_tok):
                f.write(f'"{i}","{i}","{kname}","{counter}","{kb:.6f}"\n')
            for i in range(n_other):
                f.write(f'"{i}","{i}","{other}","{counter}","1.000000"\n')
            f.write(f'"9","9","{kname}","SOMETHING_ELSE","5.0"\n')

    fp, wp = tmp_path / "fetch.csv", tmp_path / "write.csv"
    write(fp, "FETCH_SIZE", 550, 12800.0, 3)   # 12800 KB x 2 (gfx950 correction) = 26 214 400 B per launch
    write(wp, "WRITE_SIZE", 450, 10400.0, 2)   # 10400 KB = 10 649 600 B per launch
    res = bench.traffic_from_counter_csvs({"FETCH_SIZE": str(fp), "WRITE_SIZE": str(wp)})
    tok = res["k_tok_lds"]
    assert tok["fetch_dispatches"] == 550 and tok["write_dispatches"] == 450
    assert tok["fetch_per_dispatch"] == 12800.0 * 2048 and tok["write_per_dispatch"] == 10400.0 * 1024
    assert tok["bytes_per_dispatch"] == 12800.0 * 2048 + 10400.0 * 1024
    # what round 5 reported for the same files: (sum fetch + sum write) / 550 -- 8.5 MB instead of 10.6 MB of writes
    wrong = (tok["fetch"] + tok["write"]) / tok["fetch_dispatches"]
    assert abs(wrong - tok["bytes_per_dispatch"]) > 1.9e6
    assert res["k_scan_blocks"]["fetch_dispatches"] == 3 and res["k_scan_blocks"]["write_dispatches"] == 2
