"""``gtars.models`` mirror: ``Region`` and ``RegionSet``.

Signature-compatible with the reference's pyo3 classes
(gtars-python/src/models/region.rs, gtars-python/src/models/region_set.rs:69-478)
for the part of the surface that sits on the overlap hot path: construction
(path / from_regions / from_vectors), iteration, and the overlap operations
``count_overlaps / any_overlaps / find_overlaps / subset_by_overlaps`` which
index ``other`` on the GPU (IndexedRegionSet::new -> AIList by default) and
query ``self``.  BED parsing + sorting is done by the C++ host layer.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterable, List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import KIND_AILIST, check, cstr_array, dec, lib, ptr, take_u32


class Region:
    """gtars.models.Region(chr, start, end, rest) -- region.rs:11-17 of gtars-python/src/models."""

    __slots__ = ("chr", "start", "end", "rest")

    def __init__(self, chr: str, start: int, end: int, rest: Optional[str] = None):
        if not (0 <= int(start) <= 0xFFFFFFFF and 0 <= int(end) <= 0xFFFFFFFF):
            raise OverflowError("start/end must fit in u32")
        self.chr = str(chr)
        self.start = int(start)
        self.end = int(end)
        self.rest = rest

    def __repr__(self) -> str:
        return f"Region -> {self.chr} {self.start} {self.end}"

    def __str__(self) -> str:
        return f"{self.chr}\t{self.start}\t{self.end}" + (f"\t{self.rest}" if self.rest is not None else "")

    def __len__(self) -> int:
        return self.end - self.start

    def __eq__(self, other) -> bool:
        if not isinstance(other, Region):
            return NotImplemented
        return self.chr == other.chr and self.start == other.start and self.end == other.end

    def __ne__(self, other) -> bool:
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    def __hash__(self):
        return hash((self.chr, self.start, self.end, self.rest))


class RegionSet:
    """gtars.models.RegionSet -- a BED file (parsed + sorted by (chr, start)) or an in-memory list."""

    def __init__(self, path):
        p = str(path)
        h = C.c_void_p()
        st = lib.gtars_regionset_from_bed(p.encode(), C.byref(h))
        if st != 0:
            # PyRegionSet::py_new maps every error to RuntimeError (region_set.rs:79-84)
            raise RuntimeError(_lib.last_error())
        self._h = h
        self.path = p
        self._strands: Optional[List[str]] = None
        self._curr = 0

    # -- alternate constructors ------------------------------------------------
    @classmethod
    def _from_handle(cls, h, strands=None) -> "RegionSet":
        self = cls.__new__(cls)
        self._h = h
        self.path = None
        self._strands = strands
        self._curr = 0
        return self

    @classmethod
    def from_regions(cls, regions: Sequence[Region], strands: Optional[Sequence[str]] = None) -> "RegionSet":
        regions = list(regions)
        if strands is not None and len(strands) != len(regions):
            raise ValueError(f"strands length ({len(strands)}) must match regions length ({len(regions)})")
        return cls._from_columns([r.chr for r in regions], [r.start for r in regions], [r.end for r in regions],
                                 [r.rest for r in regions], strands)

    @classmethod
    def from_vectors(cls, chrs: Sequence[str], starts: Sequence[int], ends: Sequence[int],
                     strands: Optional[Sequence[str]] = None) -> "RegionSet":
        if len(starts) != len(chrs) or len(ends) != len(chrs):
            raise ValueError("chrs, starts, and ends must have the same length")
        if strands is not None and len(strands) != len(chrs):
            raise ValueError(f"strands length ({len(strands)}) must match regions length ({len(chrs)})")
        return cls._from_columns(list(chrs), starts, ends, None, strands)

    @classmethod
    def _from_columns(cls, chrs, starts, ends, rest, strands) -> "RegionSet":
        n = len(chrs)
        s = np.ascontiguousarray(starts, dtype=np.uint32)
        e = np.ascontiguousarray(ends, dtype=np.uint32)
        carr, _keep1 = cstr_array(chrs)
        rarr, _keep2 = (cstr_array(rest) if rest is not None else (None, None))
        h = C.c_void_p()
        check(lib.gtars_regionset_from_arrays(C.cast(carr, C.c_void_p), ptr(s), ptr(e),
                                              C.cast(rarr, C.c_void_p) if rarr is not None else None, n, C.byref(h)))
        return cls._from_handle(h, list(strands) if strands is not None else None)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib.gtars_regionset_free(self._h)
                self._h = None
        except Exception:
            pass

    # -- columns -----------------------------------------------------------------
    def __len__(self) -> int:
        return int(lib.gtars_regionset_len(self._h))

    def _col(self, fn) -> np.ndarray:
        n = len(self)
        if n == 0:
            return np.zeros(0, dtype=np.uint32)
        p = fn(self._h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n,)).copy()

    @property
    def starts(self) -> np.ndarray:
        return self._col(lib.gtars_regionset_starts)

    @property
    def ends(self) -> np.ndarray:
        return self._col(lib.gtars_regionset_ends)

    @property
    def chrom_names(self) -> List[str]:
        return [dec(lib.gtars_regionset_chrom_name(self._h, i)) for i in range(lib.gtars_regionset_n_chrom(self._h))]

    @property
    def chrom_ids(self) -> np.ndarray:
        return self._col(lib.gtars_regionset_chrom_ids)

    @property
    def header(self) -> Optional[str]:
        return dec(lib.gtars_regionset_header(self._h))

    @property
    def strands(self) -> List[str]:
        return list(self._strands) if self._strands is not None else ["*"] * len(self)

    def __getitem__(self, i: int) -> Region:
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("Index out of bounds")
        names = self.chrom_names
        c = int(self.chrom_ids[i])
        return Region(names[c], int(self.starts[i]), int(self.ends[i]), dec(lib.gtars_regionset_rest(self._h, i)))

    @property
    def regions(self) -> List[Region]:
        names, ids, s, e = self.chrom_names, self.chrom_ids, self.starts, self.ends
        return [Region(names[int(ids[i])], int(s[i]), int(e[i]), dec(lib.gtars_regionset_rest(self._h, i)))
                for i in range(len(self))]

    def __iter__(self):
        return iter(self.regions)

    def __repr__(self) -> str:
        return f"RegionSet with {len(self)} regions."

    # -- overlap operations (gtars-python/src/models/region_set.rs:445-478) ---------
    def count_overlaps(self, other: "RegionSet") -> List[int]:
        out = np.zeros(len(self), dtype=np.uint32)
        check(lib.gtars_regionset_count_overlaps(self._h, other._h, KIND_AILIST, 0, 0, ptr(out)))
        return [int(x) for x in out]

    def any_overlaps(self, other: "RegionSet") -> List[bool]:
        out = np.zeros(len(self), dtype=np.uint8)
        check(lib.gtars_regionset_any_overlaps(self._h, other._h, KIND_AILIST, 0, 0, ptr(out)))
        return [bool(x) for x in out]

    def find_overlaps(self, other: "RegionSet") -> List[List[int]]:
        n = len(self)
        offsets = np.zeros(n + 1, dtype=np.uint64)
        p, cnt = C.c_void_p(), C.c_uint64()
        check(lib.gtars_regionset_find_overlaps(self._h, other._h, KIND_AILIST, 0, 0, ptr(offsets), C.byref(p), C.byref(cnt)))
        idx = take_u32(p, cnt.value)
        return [[int(v) for v in idx[int(offsets[i]):int(offsets[i + 1])]] for i in range(n)]

    def subset_by_overlaps(self, other: "RegionSet") -> "RegionSet":
        counts = self.count_overlaps(other)
        regs = self.regions
        return RegionSet.from_regions([r for r, c in zip(regs, counts) if c > 0])
