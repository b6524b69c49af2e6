"""Deterministic hg38-shaped synthetic BED data.

All randomness comes from splitmix64 streams, vectorised with numpy: the k-th
output of a stream seeded with ``s`` is ``mix(s + (k+1) * GAMMA)``, so whole
arrays can be generated at once and a C++ generator with the same recurrence
produces identical data.  Used by ``bench.py`` and the parity tests; nothing
here reads the reference checkout.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

# tests/hg38.chrom.sizes of the reference (25 chromosomes, 3,088,286,401 bp)
HG38 = [
    ("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555), ("chr5", 181538259),
    ("chr6", 170805979), ("chr7", 159345973), ("chr8", 145138636), ("chr9", 138394717), ("chr10", 133797422),
    ("chr11", 135086622), ("chr12", 133275309), ("chr13", 114364328), ("chr14", 107043718),
    ("chr15", 101991189), ("chr16", 90338345), ("chr17", 83257441), ("chr18", 80373285), ("chr19", 58617616),
    ("chr20", 64444167), ("chr21", 46709983), ("chr22", 50818468), ("chrX", 156040895), ("chrY", 57227415),
    ("chrM", 16569),
]
CHROM_NAMES = [n for n, _ in HG38]
CHROM_SIZES = np.array([s for _, s in HG38], dtype=np.int64)
N_CHROM = len(HG38)
UNKNOWN_CHROM = 0xFFFFFFFF

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix_stream(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """outputs offset .. offset+n-1 of the splitmix64 stream seeded with ``seed``."""
    with np.errstate(over="ignore"):
        k = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + k * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _uniform(seed: int, n: int, hi, offset: int = 0) -> np.ndarray:
    """next() % hi, elementwise (hi scalar or array) -> int64."""
    return (splitmix_stream(seed, n, offset) % np.asarray(hi, dtype=np.uint64)).astype(np.int64)


def make_universe(n_regions: int = 100_000, seed: int = 3, overlapping: bool = False) -> Dict[str, np.ndarray]:
    """C2 universe: consensus-peak-like, non-overlapping, karyotype order (ids ascend along the genome).

    ``overlapping=True`` is the C2' "ChIP-like" variant: no clipping and 1 % wide intervals U[5e3, 1e5).
    Returns dict(chrom u32, start u32, end u32); id of a region == its row.
    """
    total = int(CHROM_SIZES.sum())
    per = (n_regions * CHROM_SIZES) // total
    per[0] += n_regions - int(per.sum())
    chroms, starts, ends = [], [], []
    off = 0
    for c in range(N_CHROM):
        n_c = int(per[c])
        if n_c == 0:
            continue
        span = max(int(CHROM_SIZES[c]) - 2000, 1)
        s = np.sort(_uniform(seed, n_c, span, off))
        w = 150 + _uniform(seed + 1000, n_c, 850, off)
        if overlapping:
            wide = _uniform(seed + 2000, n_c, 100, off) == 0
            w = np.where(wide, 5000 + _uniform(seed + 3000, n_c, 95000, off), w)
            e = s + w
        else:
            e = s + w
            e[:-1] = np.minimum(e[:-1], s[1:])
            keep = e > s
            s, e = s[keep], e[keep]
        off += n_c
        chroms.append(np.full(len(s), c, dtype=np.uint32))
        starts.append(s.astype(np.uint32))
        ends.append(e.astype(np.uint32))
    return {
        "chrom": np.concatenate(chroms),
        "start": np.concatenate(starts),
        "end": np.concatenate(ends),
    }


def make_queries(universe: Dict[str, np.ndarray], n_queries: int = 1_000_000, seed: int = 4,
                 unknown_per_mille: int = 1) -> Dict[str, np.ndarray]:
    """C2 queries: 70 % near a universe region, 30 % background, 0.1 % on an unknown chromosome, shuffled."""
    nu = len(universe["chrom"])
    kind = _uniform(seed, n_queries, 1000)
    is_unknown = kind < unknown_per_mille
    is_signal = (~is_unknown) & (kind < 700)
    width = 50 + _uniform(seed + 1, n_queries, 550)
    # signal
    u = _uniform(seed + 2, n_queries, max(nu, 1))
    us = universe["start"][u].astype(np.int64)
    ue = universe["end"][u].astype(np.int64)
    uw = np.maximum(ue - us, 1)
    jitter = _uniform(seed + 3, n_queries, 2 * uw + 1) - uw
    mid = (us + ue) // 2 + jitter
    sig_start = np.maximum(mid - width // 2, 0)
    sig_chrom = universe["chrom"][u].astype(np.int64)
    # background: chromosome proportional to length, uniform start
    total = int(CHROM_SIZES.sum())
    g = _uniform(seed + 4, n_queries, total)
    cum = np.cumsum(CHROM_SIZES)
    bg_chrom = np.searchsorted(cum, g, side="right")
    bg_start = g - (cum[bg_chrom] - CHROM_SIZES[bg_chrom])
    chrom = np.where(is_signal, sig_chrom, bg_chrom)
    start = np.where(is_signal, sig_start, bg_start)
    end = start + width
    chrom = np.where(is_unknown, UNKNOWN_CHROM, chrom)
    # shuffle: order by a random key (deterministic)
    order = np.argsort(splitmix_stream(seed + 5, n_queries), kind="stable")
    return {
        "chrom": chrom[order].astype(np.uint32),
        "start": start[order].astype(np.uint32),
        "end": end[order].astype(np.uint32),
    }


def make_igd_db(n_intervals: int, n_files: int, seed: int = 6, offset: int = 0) -> Dict[str, np.ndarray]:
    """C3 database: ChIP-like widths 200+U[0,800), hg38-shaped, file uniform.  ``offset``: rows offset .. offset+n-1 of
    the same database (the streams are counter-based, so a database can be produced in pieces)."""
    total = int(CHROM_SIZES.sum())
    g = _uniform(seed, n_intervals, total, offset)
    cum = np.cumsum(CHROM_SIZES)
    chrom = np.searchsorted(cum, g, side="right")
    start = g - (cum[chrom] - CHROM_SIZES[chrom])
    width = 200 + _uniform(seed + 1, n_intervals, 800, offset)
    return {
        "chrom": chrom.astype(np.uint32),
        "start": start.astype(np.int32),
        "end": (start + width).astype(np.int32),
        "file": _uniform(seed + 2, n_intervals, n_files, offset).astype(np.uint32),
    }


def igd_db_chunks(n_intervals: int, n_files: int, seed: int = 6, chunk: int = 4_000_000):
    """make_igd_db(n_intervals, ...) as a sequence of row chunks (concatenated: the same database, same order): what a
    rank that owns a few chromosomes ingests without ever holding the other chromosomes' rows (sharding.ShardedIgd)."""
    def chunks():
        for lo in range(0, n_intervals, chunk):
            yield make_igd_db(min(chunk, n_intervals - lo), n_files, seed, lo)
    return chunks


def make_background_queries(n_queries: int, seed: int = 7) -> Dict[str, np.ndarray]:
    """C3 queries: the C2 background law (chromosome proportional to length, width 50+U[0,550))."""
    total = int(CHROM_SIZES.sum())
    g = _uniform(seed, n_queries, total)
    cum = np.cumsum(CHROM_SIZES)
    chrom = np.searchsorted(cum, g, side="right")
    start = g - (cum[chrom] - CHROM_SIZES[chrom])
    width = 50 + _uniform(seed + 1, n_queries, 550)
    return {"chrom": chrom.astype(np.uint32), "start": start.astype(np.uint32), "end": (start + width).astype(np.uint32)}


def make_single_chrom(n: int, seed: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """C1: n intervals on chr1, start ~ U[0,1e6), width ~ U[50,500]."""
    s = _uniform(seed, n, 1_000_000)
    w = 50 + _uniform(seed + 1, n, 451)
    return np.zeros(n, dtype=np.uint32), s.astype(np.uint32), (s + w).astype(np.uint32)


def write_config5_inputs(tmp: str, universe: Dict[str, np.ndarray], files: int, frags: int, clusters: int = 20, barcodes: int = 500):
    """BASELINE config 5 inputs under ``tmp``: ``universe.bed``, ``frags/sample<k>.bed.gz`` (``frags`` position-sorted
    fragments and ``barcodes`` barcodes each -- SURVEY 8d C5: 500 --, gzip level 1) and ``map.tsv`` (80 % of the barcodes mapped
    to ``clusters`` clusters: the rest stands for cells dropped in QC).  -> (universe path, fragment dir, map path, compressed
    bytes)."""
    import gzip
    import os

    names = CHROM_NAMES
    ub = os.path.join(tmp, "universe.bed")
    with open(ub, "w") as fh:
        fh.write("".join(f"{names[c]}\t{s}\t{e}\n" for c, s, e in zip(universe["chrom"], universe["start"], universe["end"])))
    fd = os.path.join(tmp, "frags")
    os.mkdir(fd)
    name_arr = np.array(names + ["chrUn_synthetic"])

    def one(k: int) -> int:
        q = make_queries(universe, frags, seed=5000 + k)
        order = np.lexsort((q["start"], q["chrom"]))
        bc = np.random.default_rng(k).integers(0, barcodes, frags)
        c = np.minimum(q["chrom"][order], len(names))
        cols = [name_arr[c], q["start"][order].astype(str), q["end"][order].astype(str),
                np.char.add("BC", np.char.zfill(bc.astype(str), 5)), np.full(frags, "1")]
        text = "\n".join("\t".join(r) for r in zip(*cols)) + "\n"
        data = gzip.compress(text.encode(), compresslevel=1)
        with open(os.path.join(fd, f"sample{k:05d}.bed.gz"), "wb") as fh:
            fh.write(data)
        return len(data)

    # (threads: the sorts, the random draws and zlib release the interpreter lock)
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=max(1, min(16, os.cpu_count() or 1))) as ex:
        total_bytes = sum(ex.map(one, range(files)))
    map_lines = [f"sample{k:05d}+BC{b:05d}\tcl{(k + b) % clusters}" for k in range(files) for b in range(barcodes * 4 // 5)]
    mp = os.path.join(tmp, "map.tsv")
    with open(mp, "w") as fh:
        fh.write("\n".join(map_lines) + "\n")
    return ub, fd, mp, total_bytes
