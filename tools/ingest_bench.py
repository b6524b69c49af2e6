#!/usr/bin/env python3
"""Host ingest rates: BED -> RegionSet and fragment file -> SoA columns."""
import ctypes as C, gzip, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gtars_amd import _lib


def main():
    n = int(os.environ.get("LINES", "2000000"))
    rng = np.random.default_rng(1)
    chroms = [f"chr{i}" for i in list(range(1, 23)) + ["X", "Y"]]
    c = rng.integers(0, 24, n); s = rng.integers(0, 100_000_000, n); e = s + rng.integers(50, 600, n)
    tmp = tempfile.mkdtemp(prefix="gtars_ingest_")
    bed = os.path.join(tmp, "big.bed")
    text = "".join(f"{chroms[a]}\t{b}\t{d}\tpeak{i}\t{i % 1000}\n" for i, (a, b, d) in enumerate(zip(c, s, e)))
    open(bed, "w").write(text)
    with gzip.open(bed + ".gz", "wt", compresslevel=6) as fh:
        fh.write(text)
    out = {"lines": n, "bed_MB": round(len(text) / 1e6, 1), "host_threads": os.cpu_count()}
    for th in ("1", "0"):
        if th == "0":
            os.environ.pop("GTARS_HOST_THREADS", None)
        else:
            os.environ["GTARS_HOST_THREADS"] = th
        _lib.lib.gtars_debug_reload_env()  # (the library snapshots its switches at first use)
        for path in (bed, bed + ".gz"):
            best = 1e9
            for _ in range(3):
                h = C.c_void_p(); t = time.perf_counter()
                assert _lib.lib.gtars_regionset_from_bed(path.encode(), C.byref(h)) == 0
                best = min(best, time.perf_counter() - t)
                _lib.lib.gtars_regionset_free(h)
            out[f"regionset_{'gz' if path.endswith('.gz') else 'plain'}_{'1thread' if th == '1' else 'allthreads'}"] = {
                "s": round(best, 3), "Mlines_per_s": round(n / best / 1e6, 2), "MB_per_s": round(len(text) / best / 1e6)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
