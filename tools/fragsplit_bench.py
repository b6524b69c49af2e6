#!/usr/bin/env python3
"""BASELINE config 5 (scaled to one GPU): "gtars-fragsplit -> tokenizer" over many scATAC fragment files.

Writes FILES synthetic fragment files (.bed.gz, FRAGS fragments and 200 barcodes each), a barcode -> cluster map
(CLUSTERS clusters, 80 % of the barcodes mapped: the rest stands for cells dropped in QC) and a 100k-region universe, then
times, end to end from the .gz files:
  two_step   pseudobulk_fragment_files (cluster_<id>.bed.gz written, gzip level 6) + tokenize_fragment_file per cluster file
  fused      fragsplit_tokenize: the same per-cluster result without the intermediate files
and the stages on their own.  Prints one JSON line."""
import gzip, json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gtars_amd
from gtars_amd import synth
from gtars_amd.fragsplit import BarcodeToClusterMap, fragsplit_tokenize, pseudobulk_fragment_files
from gtars_amd.tokenizers import Tokenizer, tokenize_fragment_files


def main():
    K = int(os.environ.get("FILES", "1000")); n = int(os.environ.get("FRAGS", "10000")); ncl = int(os.environ.get("CLUSTERS", "20"))
    tmp = tempfile.mkdtemp(prefix="gtars_fragsplit_")
    try:
        u = synth.make_universe(100_000)
        t = time.time()
        ub, fd, mp, total_bytes = synth.write_config5_inputs(tmp, u, K, n, ncl)
        t_gen = time.time() - t
        tok = Tokenizer.from_bed(ub)
        m = BarcodeToClusterMap.from_file(mp)
        out = {"files": K, "fragments_per_file": n, "fragments": K * n, "clusters": m.n_clusters(), "map_entries": len(m),
               "input_gz_MB": round(total_bytes / 1e6, 1), "gen_s": round(t_gen, 1), "host_threads": os.cpu_count()}
        od = os.path.join(tmp, "out")
        t = time.perf_counter(); st = pseudobulk_fragment_files(fd, m, od); t_split = time.perf_counter() - t
        cluster_files = [os.path.join(od, f"cluster_{l}.bed.gz") for l in m.cluster_labels()]
        t = time.perf_counter(); res2 = tokenize_fragment_files(cluster_files, tok, workers=16); t_tok = time.perf_counter() - t
        fragsplit_tokenize(fd, m, tok, as_arrays=True)  # warm-up (device buffers)
        t = time.perf_counter(); fused = fragsplit_tokenize(fd, m, tok, as_arrays=True); t_fused = time.perf_counter() - t
        # where the fused call's time goes on the Python side: the C call alone, then the conversion of its result
        import ctypes as C
        from gtars_amd import _lib
        from gtars_amd.fragsplit import _collect_cluster_results
        o_, nr_ = C.POINTER(C.POINTER(_lib.FragmentTokens))(), C.c_uint64()
        t = time.perf_counter()
        _lib.check(_lib.lib.gtars_fragsplit_tokenize(tok._h, os.fspath(fd).encode(), m._h, C.byref(o_), C.byref(nr_)))
        t_c = time.perf_counter() - t
        t = time.perf_counter(); _collect_cluster_results(o_, m, True); t_py = time.perf_counter() - t
        out["fused_call_split"] = {"c_call_s": round(t_c, 4), "python_result_conversion_s": round(t_py, 4)}
        ids_two = sum(sum(len(v) for v in d.values()) for d in res2)
        ids_fused = sum(int(v[1][-1]) for v in fused.values())
        out.update({"routed_fragments": st["written"], "token_ids": ids_fused, "same_id_count": ids_two == ids_fused,
                    "two_step": {"fragsplit_s": round(t_split, 3), "tokenize_cluster_files_s": round(t_tok, 3),
                                 "fragments_per_s": round(K * n / (t_split + t_tok))},
                    "fused": {"s": round(t_fused, 3), "fragments_per_s": round(K * n / t_fused)}})
        print(json.dumps(out), flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
