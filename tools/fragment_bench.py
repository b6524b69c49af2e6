#!/usr/bin/env python3
"""BASELINE config 5 (scaled down): fragment files -> tokenizer, end to end, one GPU.

Writes K synthetic fragment files (1e5 fragments, 500 barcodes each; .tsv.gz and .tsv), a 100k-region
universe BED, and times gtars_tokenizer_tokenize_fragment_file per file (gunzip + parse + H2D + K3 + D2H +
per-barcode grouping) next to the parse alone (gtars_fragments_read).  Reports fragments/s."""
import ctypes as C, gzip, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gtars_amd
from gtars_amd import _lib, synth
from gtars_amd.tokenizers import Tokenizer


def main():
    K = int(os.environ.get("FILES", "16")); n = int(os.environ.get("FRAGS", "100000"))
    names = synth.CHROM_NAMES
    tmp = tempfile.mkdtemp(prefix="gtars_frag_")
    u = synth.make_universe(100_000)
    ub = os.path.join(tmp, "universe.bed")
    with open(ub, "w") as fh:
        for c, s, e in zip(u["chrom"], u["start"], u["end"]):
            fh.write(f"{names[c]}\t{s}\t{e}\n")
    files = []
    for k in range(K):
        q = synth.make_queries(u, n, seed=1000 + k)
        order = np.lexsort((q["start"], q["chrom"]))
        rng = np.random.default_rng(k)
        bc = rng.integers(0, 500, n)
        text = "".join(f"{names[c] if c < len(names) else 'chrUn_synthetic'}\t{s}\t{e}\tBC{b:05d}-1\t1\n"
                       for c, s, e, b in zip(q["chrom"][order], q["start"][order], q["end"][order], bc))
        p = os.path.join(tmp, f"frag{k}.tsv")
        open(p, "w").write(text)
        with gzip.open(p + ".gz", "wt", compresslevel=6) as fh:
            fh.write(text)
        files.append(p)
    t = time.time(); tok = Tokenizer.from_bed(ub); t_build = time.time() - t
    out = {"files": K, "fragments_per_file": n, "tokenizer_build_s": round(t_build, 3),
           "host_threads": os.cpu_count(), "plain_MB": round(os.path.getsize(files[0]) / 1e6, 2),
           "gz_MB": round(os.path.getsize(files[0] + ".gz") / 1e6, 2)}
    lib = _lib.lib
    for ext in (".gz", ""):
        # warm-up
        h = C.POINTER(_lib.FragmentTokens)()
        assert lib.gtars_tokenizer_tokenize_fragment_file(tok._h, (files[0] + ext).encode(), C.byref(h)) == 0
        lib.gtars_fragment_tokens_free(h)
        t = time.perf_counter()
        for p in files:
            f = C.c_void_p()
            assert lib.gtars_fragments_read((p + ext).encode(), C.byref(f)) == 0
            lib.gtars_fragments_free(f)
        t_parse = time.perf_counter() - t
        t = time.perf_counter()
        total_ids = 0
        for p in files:
            h = C.POINTER(_lib.FragmentTokens)()
            assert lib.gtars_tokenizer_tokenize_fragment_file(tok._h, (p + ext).encode(), C.byref(h)) == 0
            total_ids += int(h.contents.offsets[h.contents.n_barcodes])
            lib.gtars_fragment_tokens_free(h)
        t_all = time.perf_counter() - t
        out["gz" if ext else "plain"] = {"read_parse_s_per_file": round(t_parse / K, 4), "end_to_end_s_per_file": round(t_all / K, 4),
                                         "fragments_per_s": round(K * n / t_all), "parse_only_fragments_per_s": round(K * n / t_parse),
                                         "ids": total_ids}
    # many files at once: one host thread per file in flight (files are independent, the GPU part is ~5 %)
    from concurrent.futures import ThreadPoolExecutor

    def one(p):
        h = C.POINTER(_lib.FragmentTokens)()
        assert lib.gtars_tokenizer_tokenize_fragment_file(tok._h, p.encode(), C.byref(h)) == 0
        k = int(h.contents.offsets[h.contents.n_barcodes])
        lib.gtars_fragment_tokens_free(h)
        return k

    os.environ["GTARS_HOST_THREADS"] = "1"  # parallelism across files, not inside one
    lib.gtars_debug_reload_env()
    many = [p + ".gz" for p in files] * 8
    for w in (8, 32, 64):
        with ThreadPoolExecutor(max_workers=w) as ex:
            t = time.perf_counter()
            ids = sum(ex.map(one, many))
            dt = time.perf_counter() - t
        out[f"gz_{w}_file_workers"] = {"fragments_per_s": round(len(many) * n / dt), "ids": ids}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
