"""CPU oracle for the gtars hot path -- TEST INFRASTRUCTURE ONLY.

This package is the checker the MI355X path is compared against.  It must only
be imported from ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; the product package ``gtars_amd`` never
imports it and fails loudly when its HIP library is missing.

Two layers:

* ``libgtars_oracle.so`` (``gtars_oracle.c``): plain-C restatement of the
  integer algorithms (Bits, AIList, multi-chromosome bucketing, tokenizer core,
  MultiChromOverlapper counts, IndexedRegionSet.find_overlaps, Igd tile walk,
  LOLA contingency).  Each function cites the reference file:line it follows.
* this module: the string world on top (BED parsing + sort, Universe/vocab,
  special tokens, TOML config, fragment files, .gtok) restated in small pure
  Python, again citing the reference.

Parity pinning: the reference is Rust-only and cannot be built or imported in
this image, so the oracle is pinned by the reference's own known-answer tests
and fixture files -- see ``tests/test_oracle_golden.py`` and ``tests/golden``.
"""
from __future__ import annotations

import ctypes as C
import gzip
import os
import struct
import subprocess
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgtars_oracle.so")

KIND_BITS = 0
KIND_AILIST = 1


def build(force: bool = False) -> str:
    """Compile the C oracle in place (gcc, a second or two)."""
    srcs = [os.path.join(_HERE, n) for n in ("gtars_oracle.c", "fragsplit_oracle.c", "gtars_oracle.h", "Makefile")]
    stale = (
        force
        or not os.path.exists(_LIB_PATH)
        or (all(os.path.exists(x) for x in srcs) and os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(x) for x in srcs))
    )
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "libgtars_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None

_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    vp, u32, u64, i32, i64 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_int64
    L.orc_index_build.restype = vp
    L.orc_index_build.argtypes = [_u32p, _u32p, _u32p, _u32p, u64, u32, C.c_int]
    L.orc_index_free.argtypes = [vp]
    L.orc_index_chrom_len.restype = u64
    L.orc_index_chrom_len.argtypes = [vp, u32]
    L.orc_index_max_len.restype = u32
    L.orc_index_max_len.argtypes = [vp, u32]
    L.orc_index_n_headers.restype = u64
    L.orc_index_n_headers.argtypes = [vp, u32]
    L.orc_index_headers.argtypes = [vp, u32, _u64p]
    L.orc_index_stored.argtypes = [vp, u32, _u32p, _u32p, _u32p]
    L.orc_find.restype = u64
    L.orc_find.argtypes = [vp, u32, u32, u32, _u32p, _u32p, _u32p, u64]
    L.orc_bits_count.restype = u64
    L.orc_bits_count.argtypes = [vp, u32, u32, u32]
    L.orc_tokenize.restype = u64
    L.orc_tokenize.argtypes = [vp, _u32p, _u32p, _u32p, u64, _u64p, _u32p, u64]
    L.orc_count_overlaps.argtypes = [vp, _u32p, _u32p, _u32p, u64, C.c_int, i32, _u64p]
    L.orc_any_overlaps.argtypes = [vp, _u32p, _u32p, _u32p, u64, C.c_int, i32, _u8p]
    L.orc_find_overlaps_regions.restype = u64
    L.orc_find_overlaps_regions.argtypes = [vp, _u32p, _u32p, _u32p, u64, C.c_int, i32, _u64p, _u32p, _u32p, _u32p, u64]
    L.orc_irs_find_overlaps.restype = u64
    L.orc_irs_find_overlaps.argtypes = [vp, _u32p, _u32p, _u32p, u64, _u32p, _u32p, _u32p, u64, C.c_int, i32, _u64p, _u64p, u64]
    L.orc_igd_new.restype = vp
    L.orc_igd_new.argtypes = [i32]
    L.orc_igd_free.argtypes = [vp]
    L.orc_igd_add.argtypes = [vp, u32, i32, i32, i32, u32]
    L.orc_igd_add_arrays.argtypes = [vp, _u32p, _i32p, _i32p, _i32p, _u32p, u64]
    L.orc_igd_finalize.argtypes = [vp]
    L.orc_igd_total_records.restype = u64
    L.orc_igd_total_records.argtypes = [vp]
    L.orc_igd_num_contigs.restype = u64
    L.orc_igd_num_contigs.argtypes = [vp]
    L.orc_igd_count_overlaps.restype = u32
    L.orc_igd_count_overlaps.argtypes = [vp, u32, i32, i32, i32, _u64p]
    L.orc_igd_count_set_overlaps.argtypes = [vp, _u32p, _u32p, _u32p, u64, i32, _u64p, u64]
    L.orc_igd_count_region_hits.argtypes = [vp, _u32p, _u32p, _u32p, u64, i32, _u64p, u64]
    L.orc_igd_find_overlaps_regionset.restype = u64
    L.orc_igd_find_overlaps_regionset.argtypes = [vp, _u32p, _u32p, _u32p, u64, i32, _u32p, _u32p, u64]
    L.orc_igd_count_overlaps_per_query.argtypes = [vp, _u32p, _u32p, _u32p, u64, i32, _u32p]
    L.orc_lola_contingency.argtypes = [_u64p, _u64p, u64, i64, i64, _i64p, _i64p, _i64p, _i64p]
    cpp = C.POINTER(C.c_char_p)
    L.orc_fragsplit_tokenize.restype = u64
    L.orc_fragsplit_tokenize.argtypes = [vp, cpp, u64, cpp, _u32p, u64, u32, cpp, u32, u32, _u64p, _u64p, _u64p]
    L.orc_splitmix64.restype = u64
    L.orc_splitmix64.argtypes = [C.POINTER(u64)]
    _lib = L
    return L


def _u32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


_EMPTY32 = np.zeros(1, dtype=np.uint32)


class Index:
    """Per-chromosome Bits / AIList collection over integer chromosome ids."""

    def __init__(self, chrom, start, end, val=None, n_chrom: Optional[int] = None, kind: int = KIND_BITS):
        chrom, start, end = _u32(chrom), _u32(start), _u32(end)
        n = len(chrom)
        if val is None:
            val = np.arange(n, dtype=np.uint32)
        val = _u32(val)
        if n_chrom is None:
            n_chrom = int(chrom.max()) + 1 if n else 0
        self.n_chrom = int(n_chrom)
        self.kind = kind
        pad = lambda a: a if n else _EMPTY32
        self._h = lib().orc_index_build(pad(chrom), pad(start), pad(end), pad(val), n, self.n_chrom, kind)

    def __del__(self):
        try:
            if self._h:
                lib().orc_index_free(self._h)
                self._h = None
        except Exception:
            pass

    def chrom_len(self, c: int) -> int:
        return int(lib().orc_index_chrom_len(self._h, c))

    def max_len(self, c: int) -> int:
        return int(lib().orc_index_max_len(self._h, c))

    def headers(self, c: int) -> List[int]:
        n = int(lib().orc_index_n_headers(self._h, c))
        out = np.zeros(max(n, 1), dtype=np.uint64)
        lib().orc_index_headers(self._h, c, out)
        return [int(x) for x in out[:n]]

    def stored(self, c: int):
        n = self.chrom_len(c)
        s, e, v = (np.zeros(max(n, 1), dtype=np.uint32) for _ in range(3))
        lib().orc_index_stored(self._h, c, s, e, v)
        return s[:n], e[:n], v[:n]

    def find(self, c: int, qs: int, qe: int):
        """Overlapper::find -> (starts, ends, vals) in reference result order."""
        cap = 64
        while True:
            s, e, v = (np.zeros(cap, dtype=np.uint32) for _ in range(3))
            n = int(lib().orc_find(self._h, c, qs, qe, s, e, v, cap))
            if n <= cap:
                return s[:n], e[:n], v[:n]
            cap = n

    def bits_count(self, c: int, qs: int, qe: int) -> int:
        return int(lib().orc_bits_count(self._h, c, qs, qe))

    def tokenize(self, qc, qs, qe) -> Tuple[np.ndarray, np.ndarray]:
        """-> (offsets u64[nq+1], ids u32[H]); no batch-level unk applied."""
        qc, qs, qe = _u32(qc), _u32(qs), _u32(qe)
        nq = len(qc)
        offsets = np.zeros(nq + 1, dtype=np.uint64)
        if nq == 0:
            return offsets, np.zeros(0, dtype=np.uint32)
        h = int(lib().orc_tokenize(self._h, qc, qs, qe, nq, offsets, _EMPTY32, 0))
        ids = np.zeros(max(h, 1), dtype=np.uint32)
        lib().orc_tokenize(self._h, qc, qs, qe, nq, offsets, ids, h)
        return offsets, ids[:h]

    def count_overlaps(self, qc, qs, qe, min_overlap: Optional[int] = None) -> np.ndarray:
        qc, qs, qe = _u32(qc), _u32(qs), _u32(qe)
        out = np.zeros(max(len(qc), 1), dtype=np.uint64)
        if len(qc):
            lib().orc_count_overlaps(self._h, qc, qs, qe, len(qc), int(min_overlap is not None), int(min_overlap or 0), out)
        return out[: len(qc)]

    def any_overlaps(self, qc, qs, qe, min_overlap: Optional[int] = None) -> np.ndarray:
        qc, qs, qe = _u32(qc), _u32(qs), _u32(qe)
        out = np.zeros(max(len(qc), 1), dtype=np.uint8)
        if len(qc):
            lib().orc_any_overlaps(self._h, qc, qs, qe, len(qc), int(min_overlap is not None), int(min_overlap or 0), out)
        return out[: len(qc)].astype(bool)

    def find_overlaps_regions(self, qc, qs, qe, min_overlap: Optional[int] = None):
        qc, qs, qe = _u32(qc), _u32(qs), _u32(qe)
        nq = len(qc)
        offsets = np.zeros(nq + 1, dtype=np.uint64)
        if nq == 0:
            z = np.zeros(0, dtype=np.uint32)
            return offsets, z, z, z
        hm, mo = int(min_overlap is not None), int(min_overlap or 0)
        h = int(lib().orc_find_overlaps_regions(self._h, qc, qs, qe, nq, hm, mo, offsets, _EMPTY32, _EMPTY32, _EMPTY32, 0))
        s, e, v = (np.zeros(max(h, 1), dtype=np.uint32) for _ in range(3))
        lib().orc_find_overlaps_regions(self._h, qc, qs, qe, nq, hm, mo, offsets, s, e, v, h)
        return offsets, s[:h], e[:h], v[:h]

    def irs_find_overlaps(self, src_chrom, src_start, src_end, qc, qs, qe, min_overlap: Optional[int] = None):
        sc, ss, se = _u32(src_chrom), _u32(src_start), _u32(src_end)
        qc, qs, qe = _u32(qc), _u32(qs), _u32(qe)
        nq = len(qc)
        offsets = np.zeros(nq + 1, dtype=np.uint64)
        if nq == 0 or len(sc) == 0:
            return offsets, np.zeros(0, dtype=np.uint64)
        hm, mo = int(min_overlap is not None), int(min_overlap or 0)
        dummy = np.zeros(1, dtype=np.uint64)
        h = int(lib().orc_irs_find_overlaps(self._h, sc, ss, se, len(sc), qc, qs, qe, nq, hm, mo, offsets, dummy, 0))
        out = np.zeros(max(h, 1), dtype=np.uint64)
        lib().orc_irs_find_overlaps(self._h, sc, ss, se, len(sc), qc, qs, qe, nq, hm, mo, offsets, out, h)
        return offsets, out[:h]


def mco_subset_by_overlaps(ix: "Index", qc, qs, qe, min_overlap: Optional[int] = None):
    """MultiChromOverlapper::subset_by_overlaps (gtars-overlaprs/src/multi_chrom_overlapper.rs:454-478): every hit of every
    query region (find_overlaps_for_region, filtered only when min_bp > 1) goes into a BTreeSet of (chr, start, end); the set
    comes back sorted and de-duplicated.  Chromosomes are integer ids here, so "sorted by chr" is by id (the string order of
    the reference is a relabelling the host layer applies).  -> (chrom, start, end) u32 arrays."""
    qc = _u32(qc)
    off, s, e, _ = ix.find_overlaps_regions(qc, qs, qe, min_overlap)
    counts = np.diff(off.astype(np.int64))
    c = np.repeat(qc, counts)
    if len(c) == 0:
        z = np.zeros(0, dtype=np.uint32)
        return z, z.copy(), z.copy()
    rows = np.unique(np.stack([c.astype(np.uint64), s.astype(np.uint64), e.astype(np.uint64)], axis=1), axis=0)
    return rows[:, 0].astype(np.uint32), rows[:, 1].astype(np.uint32), rows[:, 2].astype(np.uint32)


def irs_subset_by_overlaps(ix: "Index", src_chrom, src_start, src_end, qc, qs, qe, min_overlap: Optional[int] = None) -> np.ndarray:
    """IndexedRegionSet::subset_by_overlaps / intersect_all (gtars-overlaprs/src/indexed_region_set.rs:201-230): the union of
    find_overlaps' per-query source indices in a BTreeSet<usize> -> ascending unique source rows."""
    _, idx = ix.irs_find_overlaps(src_chrom, src_start, src_end, qc, qs, qe, min_overlap)
    return np.unique(np.asarray(idx, dtype=np.uint64)).astype(np.uint32)


class Igd:
    """Igd over integer chromosome ids (gtars-igd/src/igd.rs)."""

    def __init__(self, nbp: int = 16384):
        self._h = lib().orc_igd_new(nbp)
        self.n_files = 0

    def __del__(self):
        try:
            if self._h:
                lib().orc_igd_free(self._h)
                self._h = None
        except Exception:
            pass

    def add(self, chrom: int, start: int, end: int, value: int, file_idx: int):
        lib().orc_igd_add(self._h, chrom, start, end, value, file_idx)
        self.n_files = max(self.n_files, file_idx + 1)

    def add_arrays(self, chrom, start, end, value, file_idx):
        """Igd::add for every row, in array order (one C loop over orc_igd_add)"""
        n = len(chrom)
        c, f = _u32(chrom), _u32(file_idx)
        s, e, v = (np.ascontiguousarray(np.asarray(x).astype(np.int32, copy=False)) for x in (start, end, value))
        lib().orc_igd_add_arrays(self._h, c, s, e, v, f, n)
        if n:
            self.n_files = max(self.n_files, int(f.max()) + 1)

    def finalize(self):
        lib().orc_igd_finalize(self._h)

    def total_records(self) -> int:
        return int(lib().orc_igd_total_records(self._h))

    def num_contigs(self) -> int:
        return int(lib().orc_igd_num_contigs(self._h))

    def count_overlaps(self, chrom: int, start: int, end: int, min_overlap: int, hits: np.ndarray) -> int:
        return int(lib().orc_igd_count_overlaps(self._h, chrom, start, end, min_overlap, hits))

    def count_set_overlaps(self, qc, qs, qe, min_overlap: int = 1, n_files: Optional[int] = None) -> np.ndarray:
        qc, qs, qe = _u32(qc), _u32(qs), _u32(qe)
        f = self.n_files if n_files is None else n_files
        hits = np.zeros(max(f, 1), dtype=np.uint64)
        if len(qc):
            lib().orc_igd_count_set_overlaps(self._h, qc, qs, qe, len(qc), min_overlap, hits, f)
        return hits[:f]

    def count_region_hits(self, qc, qs, qe, min_overlap: int = 1, n_files: Optional[int] = None) -> np.ndarray:
        qc, qs, qe = _u32(qc), _u32(qs), _u32(qe)
        f = self.n_files if n_files is None else n_files
        hits = np.zeros(max(f, 1), dtype=np.uint64)
        if len(qc):
            lib().orc_igd_count_region_hits(self._h, qc, qs, qe, len(qc), min_overlap, hits, f)
        return hits[:f]

    def find_overlaps_regionset(self, qc, qs, qe, min_overlap: int = 1):
        qc, qs, qe = _u32(qc), _u32(qs), _u32(qe)
        if len(qc) == 0:
            z = np.zeros(0, dtype=np.uint32)
            return z, z
        n = int(lib().orc_igd_find_overlaps_regionset(self._h, qc, qs, qe, len(qc), min_overlap, _EMPTY32, _EMPTY32, 0))
        oq, os_ = np.zeros(max(n, 1), dtype=np.uint32), np.zeros(max(n, 1), dtype=np.uint32)
        lib().orc_igd_find_overlaps_regionset(self._h, qc, qs, qe, len(qc), min_overlap, oq, os_, n)
        return oq[:n], os_[:n]

    def count_overlaps_per_query(self, qc, qs, qe, min_overlap: int = 1) -> np.ndarray:
        qc, qs, qe = _u32(qc), _u32(qs), _u32(qe)
        out = np.zeros(max(len(qc), 1), dtype=np.uint32)
        if len(qc):
            lib().orc_igd_count_overlaps_per_query(self._h, qc, qs, qe, len(qc), min_overlap, out)
        return out[: len(qc)]


def lola_contingency(user_hits, universe_hits, user_size: int, universe_size: int):
    uh = np.ascontiguousarray(user_hits, dtype=np.uint64)
    vh = np.ascontiguousarray(universe_hits, dtype=np.uint64)
    f = len(uh)
    a, b, c, d = (np.zeros(max(f, 1), dtype=np.int64) for _ in range(4))
    if f:
        lib().orc_lola_contingency(uh, vh, f, user_size, universe_size, a, b, c, d)
    return a[:f], b[:f], c[:f], d[:f]


class SplitMix64:
    """splitmix64 stream; identical to orc_splitmix64."""

    MASK = (1 << 64) - 1

    def __init__(self, seed: int):
        self.s = seed & self.MASK

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & self.MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & self.MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & self.MASK
        return z ^ (z >> 31)


# ---------------------------------------------------------------------------
# string world (pure Python restatements; small inputs only)
# ---------------------------------------------------------------------------


def _open_text(path: str):
    """get_dynamic_reader (gtars-core/src/utils.rs:115-126): gz iff extension == 'gz'."""
    if path.endswith(".gz"):
        return gzip.open(path, "rt")  # Python's gzip reads concatenated members, like MultiGzDecoder
    return open(path, "rt")


def _rust_lines(f) -> Iterable[str]:
    """BufRead::lines(): split on \\n, strip one trailing \\r."""
    for line in f:
        if line.endswith("\n"):
            line = line[:-1]
            if line.endswith("\r"):
                line = line[:-1]
        yield line


def _parse_u32(s: str) -> Optional[int]:
    """str::parse::<u32>(): optional leading '+', ASCII digits only, must fit."""
    t = s[1:] if s.startswith("+") else s
    if not t or not t.isascii() or not t.isdigit():
        return None
    v = int(t)
    return v if v <= 0xFFFFFFFF else None


def _parse_i32(s: str) -> Optional[int]:
    t = s[1:] if s[:1] in "+-" else s
    if not t or not t.isascii() or not t.isdigit():
        return None
    v = int(s)
    return v if -(1 << 31) <= v < (1 << 31) else None


class RegionSetError(ValueError):
    pass


def read_region_set(path: str, sort: bool = True):
    """RegionSet::try_from(&Path) (gtars-core/src/models/region_set.rs:52-186)
    -> list of (chr, start, end, rest) -- sorted like RegionSet::sort (:502-505)."""
    if not os.path.isfile(path):
        raise RegionSetError(f"not a file: {path}")
    regions = []
    first_line = True
    with _open_text(path) as f:
        for line in _rust_lines(f):
            parts = line.split("\t")
            if line.startswith("browser") or line.startswith("track") or line.startswith("#"):
                first_line = False
                continue
            if first_line:
                if len(parts) >= 3 and _parse_u32(parts[1]) is None:
                    first_line = False
                    continue
                first_line = False
            if len(parts) < 3:
                raise RegionSetError(f"Error in parsing start position: {parts}")
            s, e = _parse_u32(parts[1]), _parse_u32(parts[2])
            if s is None:
                raise RegionSetError(f"Error in parsing start position: {parts}")
            if e is None:
                raise RegionSetError(f"Error in parsing end position: {parts}")
            rest = "\t".join(parts[3:])
            regions.append((parts[0], s, e, rest if rest else None))
    if not regions:
        raise RegionSetError(f"EmptyRegionSet: {path}")
    if sort:
        # stable sort by (chr bytes, start); Rust String Ord is bytewise
        regions.sort(key=lambda r: (r[0].encode("utf-8"), r[1]))
    return regions


DEFAULT_SPECIALS = ["<unk>", "<pad>", "<mask>", "<cls>", "<eos>", "<bos>", "<sep>"]
_SPECIAL_SLOTS = ["unk", "pad", "mask", "cls", "eos", "bos", "sep"]  # special_tokens.rs:59-71 order


class UniverseError(ValueError):
    pass


class Universe:
    """gtars-tokenizers/src/universe/mod.rs:35-197."""

    def __init__(self, path: str):
        with _open_text(path) as f:
            lines = list(_rust_lines(f))
        if not lines:
            raise UniverseError("UnknownUniverseType")
        first = lines[0]
        # UniverseFileType::from (universe/utils.rs:7-19)
        if first.startswith("track"):
            raise UniverseError("UnknownUniverseType")
        nparts = len(first.split("\t"))
        regions: List[str] = []
        self.names: Optional[Dict[str, str]] = None
        self.scores: Optional[Dict[str, float]] = None
        if nparts == 3:
            for line in lines:
                parts = line.split()
                if len(parts) != 3:
                    raise UniverseError(f"Error parsing line: {line}")
                regions.append(f"{parts[0]}:{parts[1]}-{parts[2]}")
        elif nparts >= 5:
            self.names, self.scores = {}, {}
            for line in lines:
                parts = line.split("\t")
                if len(parts) < 5:
                    raise UniverseError(f"Error parsing line: {line}")
                region = f"{parts[0]}:{parts[1]}-{parts[2]}"
                regions.append(region)
                self.names[region] = parts[3]
                self.scores[region] = float(parts[4].strip())
        else:
            raise UniverseError("UnknownUniverseType")
        self.regions = regions
        # generate_region_string_to_id_map (gtars-core/src/utils.rs:240-252)
        self.region_to_id: Dict[str, int] = {}
        for r in regions:
            if r not in self.region_to_id:
                self.region_to_id[r] = len(self.region_to_id)
        # generate_id_to_region_string_map (gtars-core/src/utils.rs:259-271):
        # current_id only advances when a new id is inserted, so id i <- regions[i]
        self.id_to_region: Dict[int, str] = {}
        current_id = 0
        for r in regions:
            if current_id not in self.id_to_region:
                self.id_to_region[current_id] = r
                current_id += 1
        self.special_tokens: Optional[List[str]] = None

    def add_token_to_universe(self, region: str):  # universe/mod.rs:51-56
        new_id = len(self.region_to_id)
        self.region_to_id[region] = new_id
        self.id_to_region[new_id] = region
        self.regions.append(region)

    def add_special_tokens(self, specials: Sequence[str]):  # universe/mod.rs:114-120
        self.special_tokens = list(specials)
        for t in specials:
            self.add_token_to_universe(t)

    def __len__(self):
        return len(self.region_to_id)


class TokenizerConfigError(ValueError):
    pass


def _input_file_type(path: str) -> str:
    """TokenizerInputFileType::from_path (gtars-tokenizers/src/config.rs:74-95)."""
    base = os.path.basename(path)
    stem, ext = os.path.splitext(base)
    if ext == ".gz":
        if os.path.splitext(stem)[1] == ".bed":
            return "bedgz"
        raise TokenizerConfigError("InvalidFileType")
    if ext == ".toml":
        return "toml"
    if ext == ".bed":
        return "bed"
    raise TokenizerConfigError("InvalidFileType")


class OracleTokenizer:
    """gtars-tokenizers/src/tokenizer.rs:36-279 over the C oracle index."""

    def __init__(self, path: str):
        ftype = _input_file_type(path)
        specials = dict(zip(_SPECIAL_SLOTS, DEFAULT_SPECIALS))
        kind = KIND_BITS
        universe_path = path
        if ftype == "toml":
            import tomli

            with open(path, "rb") as f:
                cfg = tomli.load(f)
            if "universe" not in cfg or not isinstance(cfg["universe"], str):
                raise TokenizerConfigError("missing universe")
            universe_path = os.path.join(os.path.dirname(path), cfg["universe"])
            for a in cfg.get("special_tokens") or []:
                if a["name"] not in specials:
                    raise TokenizerConfigError(f"bad special token name {a['name']}")
                specials[a["name"]] = a["token"]
            tt = cfg.get("tokenizer_type")
            if tt is not None:
                if tt == "bits":
                    kind = KIND_BITS
                elif tt == "ailist":
                    kind = KIND_AILIST
                else:
                    raise TokenizerConfigError(f"unknown tokenizer_type {tt}")
        self.special = specials
        self.kind = kind
        self.universe = Universe(universe_path)
        self.universe.add_special_tokens([specials[k] for k in _SPECIAL_SLOTS])
        # create_tokenize_core_from_universe (utils/mod.rs:49-99)
        self.chrom_ids: Dict[str, int] = {}
        ch, st, en, va = [], [], [], []
        for region in self.universe.regions:
            if region in self.universe.special_tokens:
                continue
            parts = region.split(":")
            se = parts[1].split("-")
            start, end = _parse_u32(se[0]), _parse_u32(se[1])
            if start is None or end is None:
                raise ValueError("unwrap on bad coordinate")
            cid = self.chrom_ids.setdefault(parts[0], len(self.chrom_ids))
            ch.append(cid)
            st.append(start)
            en.append(end)
            va.append(self.universe.region_to_id[region])
        self.index = Index(ch, st, en, va, n_chrom=len(self.chrom_ids), kind=kind)

    UNKNOWN_CHROM = 0xFFFFFFFF

    def _encode_regions(self, regions):
        qc = [self.chrom_ids.get(r[0], self.UNKNOWN_CHROM) for r in regions]
        return qc, [r[1] for r in regions], [r[2] for r in regions]

    def encode_regions(self, regions) -> List[int]:
        """Tokenizer::encode (tokenizer.rs:165-171) on (chr,start,end) tuples."""
        return [self.universe.region_to_id[t] for t in self.tokenize(regions)]

    def tokenize(self, regions) -> List[str]:
        """Tokenizer::tokenize (tokenizer.rs:140-163)."""
        qc, qs, qe = self._encode_regions(regions)
        _, ids = self.index.tokenize(qc, qs, qe)
        if len(ids) == 0:
            return [self.special["unk"]]
        return [self.universe.id_to_region[int(i)] for i in ids]

    def tokenize_path(self, path: str) -> List[str]:
        return self.tokenize(read_region_set(path))

    @property
    def vocab_size(self) -> int:
        return len(self.universe)

    def token_to_id(self, token: str) -> Optional[int]:
        return self.universe.region_to_id.get(token)

    def tokenize_fragment_file(self, path: str) -> Dict[str, List[int]]:
        """tokenize_fragment_file (utils/fragments.rs:61-82)."""
        res: Dict[str, List[int]] = {}
        with _open_text(path) as f:
            for i, line in enumerate(_rust_lines(f)):
                if line.startswith("#"):
                    continue
                parts = line.split()
                if len(parts) < 5:
                    raise ValueError(f"Invalid fragment file detected at line: {i}")
                s, e = _parse_u32(parts[1]), _parse_u32(parts[2])
                if s is None or e is None:
                    raise ValueError(f"Failed to parse position at line {i}")
                ids = self.encode_regions([(parts[0], s, e)])
                res.setdefault(parts[3], []).extend(ids)
        return res


# --------------------------------------------------------------- IGD front


def igd_parse_bed_line(line: str):
    """Igd::parse_bed_line (gtars-igd/src/igd.rs:850-867)."""
    fields = line.split("\t")
    if len(fields) < 3:
        return None
    chrom = fields[0]
    start, end = _parse_i32(fields[1]), _parse_i32(fields[2])
    if start is None or end is None:
        return None
    if len(chrom.encode()) >= 40 or end <= 0:
        return None
    score = -1
    if len(fields) >= 5:
        v = _parse_i32(fields[4])
        score = v if v is not None else -1
    return chrom, start, end, score


class OracleIgdDb:
    """Igd::from_bed_files / from_bed_dir (igd.rs:170-242) with string chroms."""

    def __init__(self, paths: Sequence[str]):
        self.chrom_ids: Dict[str, int] = {}
        self.igd = Igd()
        self.file_info: List[Tuple[str, int, float]] = []
        for p in paths:
            try:
                f = _open_text(p)
            except OSError:
                continue
            count, total_width, has_valid = 0, 0, False
            file_idx = len(self.file_info)
            with f:
                for line in _rust_lines(f):
                    rec = igd_parse_bed_line(line)
                    if rec is None:
                        continue
                    has_valid = True
                    chrom, start, end, score = rec
                    if start >= 0:
                        cid = self.chrom_ids.setdefault(chrom, len(self.chrom_ids))
                        self.igd.add(cid, start, end, score, file_idx)
                        count += 1
                        total_width += end - start
            if not has_valid:
                continue
            self.file_info.append((os.path.basename(p), count, total_width / count if count else 0.0))
        self.igd.n_files = len(self.file_info)
        self.igd.finalize()

    @classmethod
    def from_bed_dir(cls, d: str) -> "OracleIgdDb":
        files = sorted(
            os.path.join(d, n)
            for n in os.listdir(d)
            if os.path.isfile(os.path.join(d, n)) and os.path.splitext(n)[1] in (".bed", ".gz")
        )
        return cls(files)

    UNKNOWN_CHROM = 0xFFFFFFFF

    def encode(self, regions):
        qc = [self.chrom_ids.get(r[0], self.UNKNOWN_CHROM) for r in regions]
        return qc, [r[1] for r in regions], [r[2] for r in regions]

    def count_set_overlaps(self, regions, min_overlap: int = 1):
        return self.igd.count_set_overlaps(*self.encode(regions), min_overlap=min_overlap, n_files=len(self.file_info))

    def count_region_hits(self, regions, min_overlap: int = 1):
        return self.igd.count_region_hits(*self.encode(regions), min_overlap=min_overlap, n_files=len(self.file_info))


# ------------------------------------------------------------------- gtok


def write_tokens_to_gtok(filename: str, tokens: Sequence[int]):
    """gtars-io/src/gtok.rs:125-165."""
    parent = os.path.dirname(filename)
    if parent:
        os.makedirs(parent, exist_ok=True)
    small = all(t <= 0xFFFF for t in tokens)
    with open(filename, "wb") as f:
        f.write(b"GTOK")
        f.write(bytes([0x01 if small else 0x02]))
        fmt = "<H" if small else "<I"
        for t in tokens:
            f.write(struct.pack(fmt, t))


def read_tokens_from_gtok(filename: str) -> List[int]:
    """gtars-io/src/gtok.rs:174-210."""
    with open(filename, "rb") as f:
        data = f.read()
    if len(data) < 5 or data[:4] != b"GTOK":
        raise ValueError("File doesn't appear to be a valid .gtok file.")
    flag = data[4]
    body = data[5:]
    if flag == 0x01:
        n = len(body) // 2
        return list(struct.unpack(f"<{n}H", body[: 2 * n]))
    if flag == 0x02:
        n = len(body) // 4
        return list(struct.unpack(f"<{n}I", body[: 4 * n]))
    raise ValueError("Invalid data format flag found in gtok file")


# ----------------------------------------------------------------------------- gtars-fragsplit
def remove_all_extensions(path: str) -> str:
    """gtars-core/src/utils.rs:372-387."""
    stem = os.path.basename(path)

    def ext(name):  # Path::extension
        i = name.rfind(".")
        return None if i <= 0 else name[i + 1:]

    if ext(stem) is not None:  # Path::file_stem
        stem = stem[: stem.rfind(".")]
    while ext(stem) is not None:
        stem = stem[: stem.rfind(".")]
    return stem


class OracleBarcodeMap:
    """BarcodeToClusterMap::from_file (gtars-fragsplit/src/map.rs:34-81)."""

    def __init__(self, path: str):
        self.map: Dict[str, str] = {}
        self.cluster_labels = set()
        with open(path, "r") as f:
            for line in _rust_lines(f):
                parts = line.split()
                if len(parts) < 2:
                    raise ValueError(f"Invalid line format: Expected two tab-separated values, found: {line!r}")
                self.map[parts[0]] = parts[1]
                self.cluster_labels.add(parts[1])

    def get_cluster_from_barcode(self, key: str) -> Optional[str]:
        return self.map.get(key)


def fragsplit(files_dir: str, mapping: OracleBarcodeMap, file_order: Optional[Sequence[str]] = None) -> Dict[str, List[str]]:
    """pseudobulk_fragment_files (gtars-fragsplit/src/split.rs:36-151) with the cluster files kept in memory:
    {cluster label: [output lines]}.  The reference visits the files in read_dir order (unspecified); the product
    visits them in byte order of their names, which is the default here too."""
    out: Dict[str, List[str]] = {c: [] for c in mapping.cluster_labels}
    names = list(file_order) if file_order is not None else sorted(
        n for n in os.listdir(files_dir) if os.path.isfile(os.path.join(files_dir, n)))
    for name in names:
        path = os.path.join(files_dir, name)
        stem = remove_all_extensions(path)
        with _open_text(path) as f:
            for index, line in enumerate(_rust_lines(f)):
                parts = line.split()
                if len(parts) < 5:
                    raise ValueError(f"Failed to parse fragments file at line {index}: {line}")
                chr_, start, end, barcode, support = parts[:5]
                cluster = mapping.get_cluster_from_barcode(f"{stem}+{barcode}")
                if cluster is not None:
                    out[cluster].append(f"{chr_}\t{start}\t{end}\t{barcode}\t{support}\n")
    return out


def fragsplit_tokenize_compiled(paths: Sequence[str], mapping: OracleBarcodeMap, tok: "OracleTokenizer") -> Dict[str, Tuple[int, int, int]]:
    """The config-5 pipeline in compiled C on one thread (fragsplit_oracle.c: split.rs:36-151 routing + fragments.rs:61-82
    per-line tokenization): {cluster label: (ids, sum of ids, distinct barcodes)}.  The same summary of `fragsplit` +
    `OracleTokenizer` is what tests/test_oracle_golden.py holds it to."""
    labels = sorted(mapping.cluster_labels)
    lid = {l: i for i, l in enumerate(labels)}
    keys = list(mapping.map)
    names = [None] * len(tok.chrom_ids)
    for name, cid in tok.chrom_ids.items():
        names[cid] = name
    arr = lambda xs: (C.c_char_p * max(len(xs), 1))(*[x.encode() for x in xs])
    nl = max(len(labels), 1)
    ids, sm, bc = (np.zeros(nl, dtype=np.uint64) for _ in range(3))
    unk = tok.universe.region_to_id[tok.special["unk"]]
    r = lib().orc_fragsplit_tokenize(tok.index._h, arr(list(paths)), len(paths), arr(keys),
                                     _u32([lid[mapping.map[k]] for k in keys]) if keys else _EMPTY32, len(keys), len(labels),
                                     arr(names), len(names), unk, ids, sm, bc)
    if r == 0xFFFFFFFFFFFFFFFF:
        raise ValueError("fragsplit_tokenize_compiled: unreadable file or malformed fragment line")
    return {l: (int(ids[i]), int(sm[i]), int(bc[i])) for i, l in enumerate(labels)}


# ----------------------------------------------------------------------------- LOLA statistics (independent restatement)
# gtars-lola computes its p-values with statrs 0.18 (Hypergeometric::sf / cdf), a third-party crate that is not part of
# the reference checkout.  The product uses scipy.stats.hypergeom; this is a THIRD, independent evaluation of the same
# published definition (sum of the hypergeometric pmf over the tail, every term from lgamma), so that the product's
# p-values are checked against something other than themselves.  Parity with statrs itself stays unpinned: the reference
# holds no numeric p-value literal, only the inequalities ported in tests/test_lola_stats_cpu.py.
def _log_choose(n: int, k: int) -> float:
    import math

    return math.lgamma(n + 1) - math.lgamma(k + 1) - math.lgamma(n - k + 1)


def hypergeom_log_pmf(k: int, n_pop: int, k_success: int, n_draws: int) -> float:
    return _log_choose(k_success, k) + _log_choose(n_pop - k_success, n_draws - k) - _log_choose(n_pop, n_draws)


def fisher_pvalue(a: int, b: int, c: int, d: int, enrichment: bool = True) -> float:
    """ContingencyTable::fisher_pvalue (gtars-lola/src/enrichment.rs:19-53): P(X >= a) or P(X <= a), X hypergeometric."""
    import math

    n_pop, k_success, n_draws = a + b + c + d, a + b, a + c
    if n_pop == 0 or k_success == 0 or n_draws == 0 or k_success > n_pop or n_draws > n_pop:
        return 1.0
    lo, hi = max(0, n_draws - (n_pop - k_success)), min(k_success, n_draws)
    if enrichment:
        if a == 0:
            return 1.0
        ks = range(max(a, lo), hi + 1)
    else:
        ks = range(lo, min(a, hi) + 1)
    logs = [hypergeom_log_pmf(k, n_pop, k_success, n_draws) for k in ks]
    if not logs:
        return 0.0
    m = max(logs)
    return min(1.0, math.exp(m) * math.fsum(math.exp(v - m) for v in logs))


def _brent_reference(f, a: float, b: float, tol: float, max_iter: int) -> float:
    """brent (gtars-lola/src/enrichment.rs:400-486), statement by statement: the root finder the reference's odds_ratio uses."""
    eps = 2.220446049250313e-16
    fa, fb = f(a), f(b)
    if abs(fa) < tol:
        return a
    if abs(fb) < tol:
        return b
    if fa * fb > 0.0:
        return (a + b) / 2.0
    c, fc = a, fa
    d = b - a
    e = d
    for _ in range(max_iter):
        if fb * fc > 0.0:
            c, fc = a, fa
            d = b - a
            e = d
        if abs(fc) < abs(fb):
            a, b, c = b, c, b
            fa, fb, fc = fb, fc, fb
        tol1 = 2.0 * eps * abs(b) + 0.5 * tol
        m = 0.5 * (c - b)
        if abs(m) <= tol1 or fb == 0.0:
            return b
        if abs(e) >= tol1 and abs(fa) > abs(fb):
            s_ = fb / fa
            if abs(a - c) < eps:
                p_, q_ = 2.0 * m * s_, 1.0 - s_
            else:
                qv, r = fa / fc, fb / fc
                p_ = s_ * (2.0 * m * qv * (qv - r) - (b - a) * (r - 1.0))
                q_ = (qv - 1.0) * (r - 1.0) * (s_ - 1.0)
            if p_ > 0.0:
                q_ = -q_
            else:
                p_ = -p_
            if 2.0 * p_ < min(3.0 * m * q_ - abs(tol1 * q_), e * q_):
                e = d
                d = p_ / q_
            else:
                d = m
                e = m
        else:
            d = m
            e = m
        a, fa = b, fb
        if abs(d) > tol1:
            b += d
        else:
            b += tol1 if m > 0.0 else -tol1
        fb = f(b)
    return b


def odds_ratio_reference(a: int, b: int, c: int, d: int) -> float:
    """ContingencyTable::odds_ratio (gtars-lola/src/enrichment.rs:62-160) as the reference computes it: log-densities of the
    central hypergeometric distribution by recurrence (:85-99), the noncentral mean with compensated sums (:101-135), Brent's
    method on [0, 1] in omega, or in 1 / omega when the estimate is above 1 (:137-159; absolute tolerance 1e-8 in that
    variable).  The product's odds_ratio (gtars_amd/lola.py) solves the same equation with its own method: this is what
    tests/test_lola_stats_cpu.py compares it with on a grid."""
    import math

    m, n, k, x = a + c, b + d, a + b, a
    lo = k - n if k > n else 0
    hi = min(k, m)
    if lo == hi:
        return float("nan")
    if x == lo:
        return 0.0
    if x == hi:
        return float("inf")
    size = hi - lo + 1
    logdc = [0.0]
    for i in range(1, size):
        y = lo + i - 1
        logdc.append(logdc[i - 1] + math.log(m - y) + math.log(k - y) - math.log(y + 1) - math.log(n - k + y + 1))

    def mean_nhyper(omega: float) -> float:
        if omega == 0.0:
            return float(lo)
        if math.isinf(omega):
            return float(hi)
        lg = math.log(omega)
        lv = [ld + (lo + i) * lg for i, ld in enumerate(logdc)]
        mx = max(lv)
        s_ = sc = ws = wc = 0.0
        for i, v in enumerate(lv):
            w = math.exp(v - mx)
            y = float(lo + i)
            yw = y * w - wc
            wt = ws + yw
            wc = (wt - ws) - yw
            ws = wt
            sw = w - sc
            st = s_ + sw
            sc = (st - s_) - sw
            s_ = st
        return ws / s_

    xf = float(x)
    mu1 = mean_nhyper(1.0)
    if abs(mu1 - xf) < 1e-12:
        return 1.0
    if mu1 > xf:
        return _brent_reference(lambda t: mean_nhyper(t) - xf, 0.0, 1.0, 1e-8, 100)
    t = _brent_reference(lambda t: mean_nhyper(1.0 / t) - xf, 2.220446049250313e-16, 1.0, 1e-8, 100)
    return 1.0 / t


def bh_qvalues(p_value_logs: Sequence[float]) -> List[float]:
    """apply_fdr_correction for ONE user set (gtars-lola/src/output.rs:35-113), results in input order."""
    n = len(p_value_logs)
    order = sorted(range(n), key=lambda i: -p_value_logs[i])  # stable, like sort_by
    p = [0.0 if p_value_logs[i] == float("inf") else 10.0 ** (-p_value_logs[i]) for i in order]
    q = [0.0] * n
    if n:
        q[n - 1] = min(p[n - 1] * n / n, 1.0)
        for i in range(n - 2, -1, -1):
            q[i] = min(min(p[i] * n / (i + 1), q[i + 1]), 1.0)
    out = [0.0] * n
    for j, i in enumerate(order):
        out[i] = q[j]
    return out


def rank_results(p_value_log: Sequence[float], odds_ratio: Sequence[float], support: Sequence[int]):
    """rank_results + assign_min_ranks_* + f64_tied (gtars-lola/src/enrichment.rs:296-394) on ONE user set's rows, statement by
    statement: one index list sorted three times by stable sorts (so a sort's ties keep the previous sort's order), min-ranks
    with ties by BIT PATTERN (NaN ties with NaN; 0.0 and -0.0 do not tie although they compare equal), then maxRnk / meanRnk.
    Returns (rnk_pv, rnk_or, rnk_sup, max_rnk, mean_rnk) in input order."""
    import functools
    import math
    import struct

    n = len(p_value_log)

    def tied(a: float, b: float) -> bool:
        if math.isnan(a) and math.isnan(b):
            return True
        return struct.pack("<d", a) == struct.pack("<d", b)

    def partial_desc(a: float, b: float) -> int:  # b.partial_cmp(&a).unwrap_or(Equal)
        return -1 if b < a else (1 if b > a else 0)

    def min_ranks(idx, val, same):
        out = [0] * n
        rank = 1
        for i, j in enumerate(idx):
            if i > 0 and not same(val[idx[i - 1]], val[j]):
                rank = i + 1
            out[j] = rank
        return out

    idx = list(range(n))
    idx.sort(key=functools.cmp_to_key(lambda x, y: partial_desc(p_value_log[x], p_value_log[y])))
    r_pv = min_ranks(idx, p_value_log, tied)

    def or_cmp(x, y):
        ra, rb = odds_ratio[x], odds_ratio[y]
        na, nb = math.isnan(ra), math.isnan(rb)
        if na and nb:
            return 0
        if na:
            return 1
        if nb:
            return -1
        return partial_desc(ra, rb)

    idx.sort(key=functools.cmp_to_key(or_cmp))
    r_or = min_ranks(idx, odds_ratio, tied)
    idx.sort(key=functools.cmp_to_key(lambda x, y: (support[y] > support[x]) - (support[y] < support[x])))
    r_sup = min_ranks(idx, support, lambda a, b: a == b)
    mx = [max(a, b, c) for a, b, c in zip(r_pv, r_or, r_sup)]
    mean = [(a + b + c) / 3.0 for a, b, c in zip(r_pv, r_or, r_sup)]
    return r_pv, r_or, r_sup, mx, mean


def lola_row_order(p_value_log: Sequence[float], mean_rnk: Sequence[float]) -> List[int]:
    """The final sort of run_lola (gtars-lola/src/enrichment.rs:285-294) over the concatenated rows: pValueLog descending, then
    meanRnk ascending, stable."""
    import functools

    def cmp(x, y):
        a, b = p_value_log[x], p_value_log[y]
        if b < a:
            return -1
        if b > a:
            return 1
        a, b = mean_rnk[x], mean_rnk[y]
        return -1 if a < b else (1 if a > b else 0)

    return sorted(range(len(p_value_log)), key=functools.cmp_to_key(cmp))


class MutableBits:
    """Bits with ``insert`` and ``seek`` -- a literal restatement on Python lists of gtars-overlaprs/src/bits.rs:
    build 101-128, insert 209-222, lower_bound 250-264, bsearch_seq_ref 304-322, seek 364-386, IterFind 433-446.
    Pure-Python loops: small cases only.  Intervals are (start, end, val) with u32 coordinates."""

    def __init__(self, intervals: Sequence[Tuple[int, int, int]]):
        self.intervals = sorted(((int(s), int(e), v) for s, e, v in intervals), key=lambda t: (t[0], t[1]))  # stable
        self.starts = sorted(t[0] for t in self.intervals)
        self.ends = sorted(t[1] for t in self.intervals)
        self.max_len = max([e - s if e >= s else 0 for s, e, _ in self.intervals], default=0)  # checked_sub -> 0

    def __len__(self) -> int:
        return len(self.intervals)

    @staticmethod
    def _bsearch_seq_ref(lt, n: int) -> int:
        """bits.rs:304-322 with ``lt(i)`` = elems[i] < key"""
        if n == 0 or not lt(0):
            return 0
        if lt(n - 1):
            return n
        cursor, length = 0, n
        while length > 1:
            half = length >> 1
            length -= half
            cursor += half if lt(cursor + half - 1) else 0
        return cursor

    def insert(self, start: int, end: int, val) -> None:
        si = self._bsearch_seq_ref(lambda i: self.starts[i] < start, len(self.starts))
        ei = self._bsearch_seq_ref(lambda i: self.ends[i] < end, len(self.ends))
        # Interval's Ord compares (start, end) (interval.rs:18-30)
        ii = self._bsearch_seq_ref(lambda i: self.intervals[i][:2] < (start, end), len(self.intervals))
        i_len = end - start if end >= start else 0
        if i_len > self.max_len:
            self.max_len = i_len
        self.starts.insert(si, start)
        self.ends.insert(ei, end)
        self.intervals.insert(ii, (start, end, val))

    def _lower_bound(self, start: int) -> int:
        size, low = len(self.intervals), 0
        while size > 0:
            half = size // 2
            other_half = size - half
            probe, other_low = low + half, low + other_half
            size = half
            low = other_low if self.intervals[probe][0] < start else low
        return low

    def _iter_from(self, off: int, start: int, stop: int) -> List[Tuple[int, int, int]]:
        out = []
        while off < len(self.intervals):
            s, e, v = self.intervals[off]
            off += 1
            if s < stop and e > start:
                out.append((s, e, v))
            elif s >= stop:
                break
        return out

    def find(self, start: int, stop: int) -> List[Tuple[int, int, int]]:
        key = start - self.max_len if start >= self.max_len else 0
        return self._iter_from(self._lower_bound(key), start, stop)

    def seek(self, start: int, stop: int, cursor: int) -> Tuple[List[Tuple[int, int, int]], int]:
        """-> (hits, updated cursor)"""
        n = len(self.intervals)
        key = start - self.max_len if start >= self.max_len else 0
        if cursor == 0 or (cursor < n and self.intervals[cursor][0] > start):
            cursor = self._lower_bound(key)
        while cursor + 1 < n and self.intervals[cursor + 1][0] < key:
            cursor += 1
        return self._iter_from(cursor, start, stop), cursor


# ------------------------------------------------------------------- CLI text front ends (gtars-cli)


def overlaprs_text(universe: str, query: str, backend: str = "bits") -> str:
    """gtars-cli/src/overlaprs/handlers.rs:21-157 restated: every line of both files is split on TAB, fields 2 and 3
    parse as u32; per chromosome Bits / AIList over the universe's intervals (val = None); per query line, in file
    order, one "chr\\tstart\\tend" line per hit in the backend's find order; unknown chromosomes are skipped."""
    def read(path):
        rows = []
        with _open_text(path) as f:
            for line in _rust_lines(f):
                fields = line.split("\t")
                if len(fields) < 3:
                    raise ValueError("Missing field")
                s, e = _parse_u32(fields[1]), _parse_u32(fields[2])
                if s is None or e is None:
                    raise ValueError("invalid digit found in string")
                rows.append((fields[0], s, e))
        return rows

    uni, qry = read(universe), read(query)
    names: Dict[str, int] = {}
    cid = [names.setdefault(c, len(names)) for c, _, _ in uni]
    kind = {"bits": KIND_BITS, "ailist": KIND_AILIST}[backend]
    ix = Index(cid, [r[1] for r in uni], [r[2] for r in uni], None, n_chrom=len(names), kind=kind)
    qc = [names.get(c, 0xFFFFFFFF) for c, _, _ in qry]
    off, hs, he, _ = ix.find_overlaps_regions(qc, [r[1] for r in qry], [r[2] for r in qry])
    out = []
    for i, (c, _, _) in enumerate(qry):
        for k in range(int(off[i]), int(off[i + 1])):
            out.append(f"{c}\t{int(hs[k])}\t{int(he[k])}\n")
    return "".join(out)


def igd_search_text(bed_paths: Sequence[str], query: str) -> str:
    """gtars-cli/src/igd/handlers.rs:74-98 restated over Igd::from_bed_files: the legacy TSV of the files with hits."""
    db = OracleIgdDb(bed_paths)
    regions = [(r[0], r[1], r[2]) for r in read_region_set(query)]  # RegionSet::try_from: parsed and sorted
    hits = db.count_set_overlaps(regions, 1)
    out = ["index\t number of regions\t number of hits\t File_name\n"]
    for i, (name, n_regions, _) in enumerate(db.file_info):
        if int(hits[i]) > 0:
            out.append(f"{i}\t{n_regions}\t{int(hits[i])}\t{name}\n")
    out.append(f"Total: {int(sum(int(h) for h in hits))}\n")
    return "".join(out)
