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
        self._rec: List[Tuple[int, int, int, int, int]] = []   # records added one by one, not yet in a chunk
        self._chunks: List[np.ndarray] = []                     # int64 [n, 5] blocks in insertion order
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

    def _flush(self) -> None:
        if self._rec:
            self._chunks.append(np.asarray(self._rec, dtype=np.int64))
            self._rec = []

    def _add_columns(self, chrom_names: Sequence[str], chrom_ids: np.ndarray, starts: np.ndarray, ends: np.ndarray,
                     values: np.ndarray, file_idx: int) -> Tuple[int, int]:
        """``add`` for whole columns (i32 coordinates, the reference's ``as i32`` casts): keeps start >= 0, end >= 0,
        start < end (igd.rs:114-116), returns (records kept, their total width)."""
        if self._engine is not None:
            raise AssertionError("Cannot add intervals after finalization")
        s = np.asarray(starts).astype(np.uint32).view(np.int32).astype(np.int64)
        e = np.asarray(ends).astype(np.uint32).view(np.int32).astype(np.int64)
        keep = (s >= 0) & (e >= 0) & (s < e)
        if not keep.any():
            return 0, 0
        self._flush()
        # dictionary ids in first-seen order of the KEPT records, like add() would assign them
        ids = np.asarray(chrom_ids)[keep]
        first = np.unique(ids, return_index=True)
        lut = np.zeros(len(chrom_names) if len(chrom_names) else 1, dtype=np.int64)
        for cid in first[0][np.argsort(first[1])]:
            lut[cid] = self._chrom_ids.setdefault(chrom_names[int(cid)], len(self._chrom_ids))
        blk = np.empty((int(keep.sum()), 5), dtype=np.int64)
        blk[:, 0] = lut[ids]
        blk[:, 1] = s[keep]
        blk[:, 2] = e[keep]
        blk[:, 3] = np.asarray(values)[keep] if np.ndim(values) else values
        blk[:, 4] = file_idx
        self._chunks.append(blk)
        return len(blk), int((blk[:, 2] - blk[:, 1]).sum())

    def finalize(self) -> None:
        """Igd::finalize (igd.rs:157-167): the records go to the device through ``gtars_igddb_from_arrays``; from here on
        the database is the same C handle the BED / .igd loaders produce."""
        if self._engine is not None:
            return
        self._flush()
        if self._chunks:
            a = np.concatenate(self._chunks) if len(self._chunks) > 1 else self._chunks[0]
            self._chunks = []
        else:
            a = np.zeros((0, 5), dtype=np.int64)
        names = [n for n, _ in sorted(self._chrom_ids.items(), key=lambda kv: kv[1])]
        c = np.ascontiguousarray(a[:, 0], dtype=np.uint32)
        st, en, va = (np.ascontiguousarray(a[:, k], dtype=np.int32) for k in (1, 2, 3))
        fi = np.ascontiguousarray(a[:, 4], dtype=np.uint32)
        narr, _k1 = cstr_array(names)
        farr, _k2 = cstr_array([f.filename for f in self.file_info])
        nreg = np.asarray([f.num_regions for f in self.file_info] or [0], dtype=np.uint32)
        avgw = np.asarray([f.avg_region_width for f in self.file_info] or [0.0], dtype=np.float64)
        h = C.c_void_p()
        check(lib.gtars_igddb_from_arrays(C.cast(narr, C.c_void_p), len(names), ptr(c), ptr(st), ptr(en), ptr(va), ptr(fi), len(c),
                                          C.cast(farr, C.c_void_p), ptr(nreg), ptr(avgw), len(self.file_info), C.byref(h)))
        self._adopt(h)

    def _adopt(self, h) -> None:
        self._db = h
        nf = int(lib.gtars_igddb_n_files(h))
        self._engine = IgdIndex.__new__(IgdIndex)
        self._engine._h = C.c_void_p(lib.gtars_igddb_engine(h))
        self._engine.n_files = int(lib.gtars_igd_n_files(self._engine._h))
        self._engine.n_chrom = int(lib.gtars_igddb_n_contigs(h))
        self._engine.close = lambda: None  # borrowed from the db handle
        self._n_files_engine = self._engine.n_files
        if not self.file_info:
            self.file_info = [FileInfo(dec(lib.gtars_igddb_file_name(h, i)), int(lib.gtars_igddb_file_num_regions(h, i)),
                                       float(lib.gtars_igddb_file_avg_width(h, i))) for i in range(nf)]

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
    def _from_db(cls, h, nbp: int = 16384) -> "Igd":
        self = cls(nbp)
        self._adopt(h)
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
            st, en = rs.starts, rs.ends
            ok = st < en  # u32 comparison first (igd.rs:296), then `as i32` and Igd::add's own rule
            count, total = self._add_columns(rs.chrom_names, rs.chrom_ids[ok], st[ok], en[ok], 0, file_idx)
            # count / width are taken over the regions with start < end (igd.rs:297-306), i32-cast widths
            s32 = st[ok].view(np.int32).astype(np.int64)
            e32 = en[ok].view(np.int32).astype(np.int64)
            n_ok = int(ok.sum())
            self.file_info.append(FileInfo(filename, n_ok, float((e32 - s32).sum()) / n_ok if n_ok else 0.0))
        self.finalize()
        return self

    @classmethod
    def from_single_region_set(cls, rs: RegionSet) -> "Igd":
        """igd.rs:609-634: value = source index, one file."""
        self = cls()
        n = len(rs)
        st, en = rs.starts.astype(np.int64), rs.ends.astype(np.int64)
        self.file_info = [FileInfo("", n, float((en - st).mean()) if n else 0.0)]
        self._add_columns(rs.chrom_names, rs.chrom_ids, rs.starts, rs.ends, np.arange(n, dtype=np.int64), 0)
        self.finalize()
        return self

    # ---- persistence: .igd + .tsv (igd.rs:320-486) ---------------------------------------
    def _chrom_names(self) -> List[str]:
        if self._db is not None:
            return [dec(lib.gtars_igddb_chrom_name(self._db, i)) for i in range(self.num_contigs())]
        return [n for n, _ in sorted(self._chrom_ids.items(), key=lambda kv: kv[1])]

    def _export(self):
        n = len(self._engine)
        c = np.zeros(n, dtype=np.uint32)
        s, e, v = (np.zeros(n, dtype=np.int32) for _ in range(3))
        f = np.zeros(n, dtype=np.uint32)
        check(lib.gtars_igd_export(self._engine._h, ptr(c), ptr(s), ptr(e), ptr(v), ptr(f)))
        return c, s, e, v, f

    def save(self, path: str) -> None:
        """Igd::save (igd.rs:425-486) through ``gtars_igddb_save``: the .igd v1 file and its companion .tsv."""
        self._require()
        check(lib.gtars_igddb_save(self._db, os.fspath(path).encode(), int(self.nbp)))

    @classmethod
    def from_igd_file(cls, path: str) -> "Igd":
        """Igd::from_igd_file (igd.rs:320-418) through ``gtars_igddb_load``.  Tile replicas are dropped on load: a record
        is kept from the tile it starts in, so the device holds every stored interval once."""
        h = C.c_void_p()
        nbp = C.c_int32()
        check(lib.gtars_igddb_load(os.fspath(path).encode(), C.byref(h), C.byref(nbp)))
        return cls._from_db(h, int(nbp.value))

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
        """igd.rs:544-561.  ``min_overlap <= 0`` is accepted like in the reference: its tile walk then also admits records
        that do not overlap the query, depending on the 16384-bp tile they fall in (igd.rs:772-846) -- reproduced here
        by the per-query kernels with the walk's tile test.  The same holds for every query method below."""
        self._require()
        return self._engine.count_set_overlaps(*self._encode(regions), min_overlap=min_overlap)[: self.num_files()]

    count_regions_overlaps = count_set_overlaps

    def count_region_hits(self, regions, min_overlap: int = 1) -> np.ndarray:
        self._require()
        return self._engine.count_region_hits(*self._encode(regions), min_overlap=min_overlap)[: self.num_files()]

    def count_region_hits_sets(self, region_sets, min_overlap: int = 1) -> np.ndarray:
        """Additive batch form: count_region_hits of every set of `region_sets` -> u64[len(region_sets), num_files].  Up to four
        sets share one pass over the database (gtars_igd_count_sets); run_lola's count step (the universe and the user sets,
        gtars-lola/src/enrichment.rs:198-221) goes through here."""
        self._require()
        return self._engine.count_sets([self._encode(r) for r in region_sets], min_overlap=min_overlap, binary=True)[:, : self.num_files()]

    def count_set_overlaps_sets(self, region_sets, min_overlap: int = 1) -> np.ndarray:
        """The pairwise counterpart: count_set_overlaps of every set -> u64[len(region_sets), num_files]."""
        self._require()
        return self._engine.count_sets([self._encode(r) for r in region_sets], min_overlap=min_overlap, binary=False)[:, : self.num_files()]

    def find_overlaps_regionset(self, query, min_overlap: int = 1) -> List[Tuple[int, int]]:
        self._require()
        q, s = self._engine.find_overlaps_regionset(*self._encode(query), min_overlap=min_overlap)
        return list(zip(q.tolist(), s.tolist()))

    def count_overlaps_per_query(self, query, min_overlap: int = 1) -> List[int]:
        self._require()
        return self._engine.count_overlaps_per_query(*self._encode(query), min_overlap=min_overlap).tolist()
