#!/usr/bin/env python3
"""Randomised differential soak of the fused fragment pipeline (gtars_fragsplit_tokenize): random folders of fragment files --
line shapes the reference accepts (CRLF, runs of blanks and tabs, leading blanks, extra columns, '+' numbers, '#' chromosomes,
unknown chromosomes, unmapped barcodes with garbage numbers, a last line without a newline), empty files, plain-text files,
gzip levels 0-9, several members per file -- under random host-side switches (threads, decoder, CRC site, pinned pool
forms, host parser), against the oracle's restatement of the two-step pipeline (split.rs:84-131 + fragments.rs:12-56).

Run on the GPU box:  python tests/soak/fuzz_fragments.py [rounds] [first seed]"""
import gzip, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gtars_amd
import oracle
from gtars_amd import _lib
from gtars_amd.fragsplit import BarcodeToClusterMap, fragsplit_tokenize, list_fragment_files
from gtars_amd.tokenizers import Tokenizer
from test_sharding_gloo import oracle_fragment_pipeline, same_cluster_results

SWITCHES = [{}, {}, {"GTARS_HOST_THREADS": "1"}, {"GTARS_HOST_THREADS": "3"}, {"GTARS_ZLIB_INFLATE": "1"}, {"GTARS_FRAG_HOST_CRC": "1"},
            {"GTARS_NO_PINNED": "1"}, {"GTARS_PINNED_POOL_MB": "0"}, {"GTARS_PINNED_MAX_MB": "1"}, {"GTARS_FRAG_HOST_PARSE": "1"}]


def one(seed, tmp, tok, otok, chroms, ub):
    rng = np.random.default_rng(seed)
    fd = os.path.join(tmp, f"frags{seed}")
    os.mkdir(fd)
    n_files = int(rng.choice([1, 2, 5, 17, 40]))
    n_bc = int(rng.choice([1, 4, 60]))
    n_cl = int(rng.choice([1, 3, 9]))
    lines_map = []
    seps = ["\t", "\t", "  ", " \t "]
    for fi in range(n_files):
        n = int(rng.choice([0, 1, 7, 500, 6000]))
        out = []
        for _ in range(n):
            kind = rng.random()
            c = chroms[int(rng.integers(0, len(chroms)))] if kind > 0.05 else ("#hdr" if kind > 0.025 else "chrZZ")
            s0 = int(rng.integers(0, 250_000))
            bc = f"B{int(rng.integers(0, n_bc + 2))}"  # (the last two are never mapped)
            sep = seps[int(rng.integers(0, len(seps)))]
            plus = "+" if rng.random() < 0.05 else ""
            line = ("  " if rng.random() < 0.05 else "") + sep.join([c, plus + str(s0), str(s0 + int(rng.integers(1, 900))), bc, "1"])
            if rng.random() < 0.1:
                line += sep + "extra" + sep + "7"
            if int(bc[1:]) >= n_bc and rng.random() < 0.3:  # unmapped barcode: the numbers are never looked at
                line = sep.join([c, "x1", "-5", bc, "?"])
            out.append(line + ("\r\n" if rng.random() < 0.1 else "\n"))
        text = "".join(out)
        if text and rng.random() < 0.2:
            text = text.rstrip("\r\n")  # a last line without a newline
        raw = text.encode()
        how = rng.random()
        if how < 0.1:
            blob = raw  # a plain-text file
        elif how < 0.3 and len(raw) > 200:
            cut = raw.rfind(b"\n", 0, len(raw) // 2) + 1
            blob = gzip.compress(raw[:cut], int(rng.integers(0, 10))) + gzip.compress(raw[cut:], int(rng.integers(0, 10))) + gzip.compress(b"")
        else:
            blob = gzip.compress(raw, int(rng.integers(0, 10)))
        with open(os.path.join(fd, f"s{fi:03d}.bed" + ("" if how < 0.1 else ".gz")), "wb") as f:  # (plain text: no ".gz")
            f.write(blob)
        lines_map += [f"s{fi:03d}+B{b}\tcl{int(rng.integers(0, n_cl))}" for b in range(n_bc) if rng.random() < 0.9]
    if not lines_map:
        lines_map = ["s000+B0\tcl0"]
    mp = os.path.join(tmp, f"map{seed}.tsv")
    with open(mp, "w") as f:
        f.write("\n".join(lines_map) + "\n")
    m = BarcodeToClusterMap.from_file(mp)
    om = oracle.OracleBarcodeMap(mp)
    want = oracle_fragment_pipeline(list_fragment_files(fd), om, otok)
    sw = SWITCHES[int(rng.integers(0, len(SWITCHES)))]
    for k, v in sw.items():
        os.environ[k] = v
    _lib.lib.gtars_debug_reload_env()
    try:
        got = fragsplit_tokenize(fd, m, tok, as_arrays=True)
        assert same_cluster_results(got, want), ("fused pipeline differs from the oracle", seed, sw)
    finally:
        for k in sw:
            os.environ.pop(k, None)
        _lib.lib.gtars_debug_reload_env()
        shutil.rmtree(fd, ignore_errors=True)
    return sum(int(v[1][-1]) for v in got.values())


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    base = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    ub = os.path.join(ROOT, "tests", "golden", "tokenizers", "peaks.bed")
    chroms = sorted({l.split()[0] for l in open(ub) if l.strip()})
    tok, otok = Tokenizer.from_bed(ub), oracle.OracleTokenizer(ub)
    tmp = tempfile.mkdtemp(prefix="gtars_fuzzfrag_")
    t = time.time()
    tot = 0
    try:
        for seed in range(rounds):
            tot += one(base + seed, tmp, tok, otok, chroms, ub)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(f"fuzz_fragments: {rounds} random folders bit-exact vs the oracle ({tot} ids compared, {time.time() - t:.0f} s)")


if __name__ == "__main__":
    main()
