"""IGD databases: the reference's ``Igd`` API (gtars-igd/src/igd.rs) over the HIP engine.

``Igd`` mirrors the Rust struct's public methods (add / finalize / from_* builders / count_* /
find_overlaps_regionset / count_overlaps_per_query).  Records are collected on the host and
uploaded at ``finalize``; all counting runs in the K5 kernels.  Queries use chromosome NAMES,
like the reference.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import UNKNOWN_CHROM, check, cstr_array, dec, lib, ptr
from .engine import IgdIndex
from .models import RegionSet


class FileInfo:
    """igd.rs:52-59"""

    def __init__(self, filename: str, num_regions: int, avg_region_width: float):
        self.filename = filename
        self.num_regions = num_regions
        self.avg_region_width = avg_region_width

    def __repr__(self):
        return f"FileInfo({self.filename!r}, {self.num_regions}, {self.avg_region_width})"


class Igd:
    """In-memory multi-file interval database (igd.rs:61-72), counted on the GPU."""

    def __init__(self, nbp: int = 16384):
        self.nbp = nbp
        self.file_info: List[FileInfo] = []
        self._chrom_ids: Dict[str, int] = {}
        self._rec: List[Tuple[int, int, int, int, int]] = []
        self._engine: Optional[IgdIndex] = None
        self._db = None  # gtars_igddb_t* when built by the C++ host from BED files

    # ---- construction (igd.rs:109-317, 609-634) -------------------------------------
    def add(self, chrom: str, start: int, end: int, value: int, file_idx: int) -> None:
        if self._engine is not None:
            raise AssertionError("Cannot add intervals after finalization")
        if start < 0 or end < 0 or start >= end:
            return
        cid = self._chrom_ids.setdefault(chrom, len(self._chrom_ids))
        self._rec.append((cid, start, end, value, file_idx))

    def finalize(self) -> None:
        if self._engine is not None:
            return
        n_files = len(self.file_info)
        if self._rec:
            a = np.asarray(self._rec, dtype=np.int64)
            n_files = max(n_files, int(a[:, 4].max()) + 1)
            self._engine = IgdIndex(a[:, 0], a[:, 1], a[:, 2], a[:, 4], a[:, 3], n_chrom=len(self._chrom_ids), n_files=n_files)
        else:
            z = np.zeros(0, dtype=np.int64)
            self._engine = IgdIndex(z, z, z, z, z, n_chrom=max(len(self._chrom_ids), 0), n_files=n_files)
        self._n_files_engine = n_files

    @classmethod
    def from_bed_files(cls, paths: Iterable[str]) -> "Igd":
        paths = [str(p) for p in paths]
        arr, _keep = cstr_array(paths)
        h = C.c_void_p()
        check(lib.gtars_igddb_from_bed_files(C.cast(arr, C.c_void_p), len(paths), C.byref(h)))
        return cls._from_db(h)

    @classmethod
    def from_bed_dir(cls, path: str) -> "Igd":
        h = C.c_void_p()
        check(lib.gtars_igddb_from_bed_dir(str(path).encode(), C.byref(h)))
        return cls._from_db(h)

    @classmethod
    def _from_db(cls, h) -> "Igd":
        self = cls()
        self._db = h
        nf = int(lib.gtars_igddb_n_files(h))
        self.file_info = [FileInfo(dec(lib.gtars_igddb_file_name(h, i)), int(lib.gtars_igddb_file_num_regions(h, i)),
                                   float(lib.gtars_igddb_file_avg_width(h, i))) for i in range(nf)]
        self._engine = IgdIndex.__new__(IgdIndex)
        self._engine._h = C.c_void_p(lib.gtars_igddb_engine(h))
        self._engine.n_files = nf
        self._engine.n_chrom = int(lib.gtars_igddb_n_contigs(h))
        self._engine.close = lambda: None  # borrowed from the db handle
        self._n_files_engine = nf
        return self

    @classmethod
    def from_region_sets(cls, sets: Iterable[Tuple[str, Sequence[Tuple[str, int, int]]]]) -> "Igd":
        """igd.rs:247-281: (filename, [(chrom, start, end)]) pairs; start >= end is skipped."""
        self = cls()
        for file_idx, (filename, regions) in enumerate(sets):
            count, total = 0, 0
            for chrom, s, e in regions:
                if s < e:
                    self.add(chrom, int(s), int(e), 0, file_idx)
                    count += 1
                    total += int(e) - int(s)
            self.file_info.append(FileInfo(filename, count, total / count if count else 0.0))
        self.finalize()
        return self

    @classmethod
    def from_named_region_sets(cls, sets: Sequence[Tuple[str, RegionSet]]) -> "Igd":
        """igd.rs:284-317"""
        self = cls()
        for file_idx, (filename, rs) in enumerate(sets):
            names, ids, st, en = rs.chrom_names, rs.chrom_ids, rs.starts, rs.ends
            count, total = 0, 0
            for i in range(len(rs)):
                if st[i] < en[i]:
                    s, e = int(np.int32(st[i])), int(np.int32(en[i]))  # `as i32`
                    self.add(names[int(ids[i])], s, e, 0, file_idx)
                    count += 1
                    total += e - s
            self.file_info.append(FileInfo(filename, count, total / count if count else 0.0))
        self.finalize()
        return self

    @classmethod
    def from_single_region_set(cls, rs: RegionSet) -> "Igd":
        """igd.rs:609-634: value = source index, one file."""
        self = cls()
        n = len(rs)
        st, en = rs.starts.astype(np.int64), rs.ends.astype(np.int64)
        self.file_info = [FileInfo("", n, float((en - st).mean()) if n else 0.0)]
        names, ids = rs.chrom_names, rs.chrom_ids
        for i in range(n):
            self.add(names[int(ids[i])], int(np.int32(rs.starts[i])), int(np.int32(rs.ends[i])), i, 0)
        self.finalize()
        return self

    def __del__(self):
        try:
            if getattr(self, "_db", None):
                lib.gtars_igddb_free(self._db)
                self._db = None
        except Exception:
            pass

    # ---- queries ---------------------------------------------------------------------
    def _require(self):
        if self._engine is None:
            raise AssertionError("Must finalize before querying")

    def _chrom_id(self, name: str) -> int:
        if self._db is not None:
            r = int(lib.gtars_igddb_chrom_id(self._db, name.encode()))
            return UNKNOWN_CHROM if r < 0 else r
        return self._chrom_ids.get(name, UNKNOWN_CHROM)

    def _encode(self, regions) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        if isinstance(regions, RegionSet):
            names = regions.chrom_names
            lut = np.asarray([self._chrom_id(n) for n in names] or [0], dtype=np.uint32)
            return lut[regions.chrom_ids] if len(regions) else np.zeros(0, np.uint32), regions.starts, regions.ends
        regions = list(regions)
        cache: Dict[str, int] = {}
        qc = np.asarray([cache.setdefault(r[0], self._chrom_id(r[0])) for r in regions], dtype=np.uint32)
        # i32 query coordinates are passed as their u32 bit patterns (the engine casts back, igd.rs:549-550)
        qs = np.asarray([r[1] for r in regions], dtype=np.int64).astype(np.uint32)
        qe = np.asarray([r[2] for r in regions], dtype=np.int64).astype(np.uint32)
        return qc, qs, qe

    def num_files(self) -> int:
        return len(self.file_info)

    def num_contigs(self) -> int:
        if self._db is not None:
            return int(lib.gtars_igddb_n_contigs(self._db))
        return len(self._chrom_ids)

    def total_records(self) -> int:
        self._require()
        return self._engine.total_records(self.nbp)

    def count_overlaps(self, chrom: str, start: int, end: int, min_overlap: int, hits: np.ndarray) -> int:
        """igd.rs:504-540: adds into ``hits`` and returns the number of overlaps of this one query."""
        self._require()
        h = self._engine.count_set_overlaps(*self._encode([(chrom, start, end)]), min_overlap=min_overlap)
        hits[: len(h)] += h.astype(hits.dtype)
        return int(h.sum())

    def count_set_overlaps(self, regions, min_overlap: int = 1) -> np.ndarray:
        self._require()
        return self._engine.count_set_overlaps(*self._encode(regions), min_overlap=min_overlap)[: self.num_files()]

    count_regions_overlaps = count_set_overlaps

    def count_region_hits(self, regions, min_overlap: int = 1) -> np.ndarray:
        self._require()
        return self._engine.count_region_hits(*self._encode(regions), min_overlap=min_overlap)[: self.num_files()]

    def find_overlaps_regionset(self, query, min_overlap: int = 1) -> List[Tuple[int, int]]:
        self._require()
        q, s = self._engine.find_overlaps_regionset(*self._encode(query), min_overlap=min_overlap)
        return list(zip(q.tolist(), s.tolist()))

    def count_overlaps_per_query(self, query, min_overlap: int = 1) -> List[int]:
        self._require()
        return self._engine.count_overlaps_per_query(*self._encode(query), min_overlap=min_overlap).tolist()
