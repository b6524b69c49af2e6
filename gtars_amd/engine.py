"""Integer-level engine objects over the C ABI: ``OverlapIndex`` and ``IgdIndex``.

These are the array fast path underneath the reference-shaped classes in
``gtars_amd.tokenizers`` / ``gtars_amd.models`` / ``gtars_amd.igd``: chromosome
names are already dictionary-encoded to dense u32 ids, coordinates are u32
arrays.  Host arrays are numpy; the ``*_device`` methods take raw device
pointers (e.g. ``tensor.data_ptr()``) and a HIP stream handle.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib
from ._lib import KIND_AILIST, KIND_BITS, as_u32, check, lib, ptr, take_u32


def _minargs(min_overlap: Optional[int]):
    return (0, 0) if min_overlap is None else (1, int(min_overlap))


class OverlapIndex:
    """Genome-wide Bits / AIList index resident in HBM.

    Mirrors ``Overlapper::build`` per chromosome (bits.rs:101-128,
    ailist.rs:105-151) behind ``MultiChromOverlapper`` bucketing
    (multi_chrom_overlapper.rs:325-351).
    """

    def __init__(self, chrom, start, end, val=None, n_chrom: Optional[int] = None, kind: int = KIND_BITS):
        chrom, start, end = as_u32(chrom), as_u32(start), as_u32(end)
        if not (len(chrom) == len(start) == len(end)):
            raise ValueError("chrom, start, end must have the same length")
        n = len(chrom)
        v = None if val is None else as_u32(val)
        if n_chrom is None:
            n_chrom = int(chrom.max()) + 1 if n else 0
        h = C.c_void_p()
        check(lib.gtars_index_build(ptr(chrom), ptr(start), ptr(end), ptr(v) if v is not None else None, n,
                                    int(n_chrom), int(kind), C.byref(h)))
        self._h = h
        self.kind = kind
        self.n_chrom = int(n_chrom)

    def close(self):
        if getattr(self, "_h", None):
            lib.gtars_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(lib.gtars_index_len(self._h))

    # -- introspection (tests) -------------------------------------------------
    def chrom_len(self, c: int) -> int:
        return int(lib.gtars_index_chrom_len(self._h, c))

    def max_len(self, c: int) -> int:
        return int(lib.gtars_index_max_len(self._h, c))

    def stored(self, c: int):
        n = self.chrom_len(c)
        s, e, v = (np.zeros(n, dtype=np.uint32) for _ in range(3))
        check(lib.gtars_index_stored(self._h, c, ptr(s), ptr(e), ptr(v)))
        return s, e, v

    def sublist_offsets(self, c: int):
        n = int(lib.gtars_index_n_sublists(self._h, c))
        if n == 0:  # (a chromosome without intervals, or a Bits-kind index)
            return []
        out = np.zeros(n, dtype=np.uint64)
        check(lib.gtars_index_sublist_offsets(self._h, c, ptr(out)))
        return [int(x) for x in out]

    # -- Bits::insert / Bits::seek (bits.rs:209-222, 364-386) -----------------------
    def insert(self, chrom: int, start: int, end: int, val: int) -> None:
        """Bits::insert: the interval goes where ``bsearch_seq_ref`` puts it (in front of equal (start, end) keys).
        The device structures are rebuilt (O(n), like the reference's Vec::insert)."""
        h = C.c_void_p()
        check(lib.gtars_index_insert(self._h, int(chrom), int(start), int(end), int(val), C.byref(h)))
        old, self._h = self._h, h
        lib.gtars_index_free(old)

    def seek(self, chrom: int, start: int, stop: int, cursor: int = 0) -> Tuple[np.ndarray, int]:
        """Bits::seek for sorted query sequences: -> (vals of the hits in stored order, updated cursor)."""
        cur, n = C.c_uint64(int(cursor)), C.c_uint64()
        out = np.empty(64, dtype=np.uint32)
        rc = lib.gtars_index_seek(self._h, int(chrom), int(start), int(stop), C.byref(cur), ptr(out), len(out), C.byref(n))
        if rc == _lib.ERR_CAPACITY:
            out = np.empty(int(n.value), dtype=np.uint32)
            cur = C.c_uint64(int(cursor))
            rc = lib.gtars_index_seek(self._h, int(chrom), int(start), int(stop), C.byref(cur), ptr(out), len(out), C.byref(n))
        check(rc)
        return out[: int(n.value)].copy(), int(cur.value)

    # -- host-array queries ----------------------------------------------------
    def tokenize(self, qc, qs, qe, out: Optional[Tuple[np.ndarray, np.ndarray]] = None) -> Tuple[np.ndarray, np.ndarray]:
        """-> (offsets u64[nq+1], ids u32[H]) in reference order; no batch-level unk.

        ``out=(offsets, ids)``: caller-provided uint64[>= nq+1] / uint32 arrays that are REUSED across calls (the
        streaming form, ``gtars_tokenize_into``: nothing is allocated, freshly mapped pages never have to be
        faulted in during the device-to-host copy); the returned arrays are views of them.  When ``ids`` is too
        small a larger array is allocated for this call."""
        qc, qs, qe = as_u32(qc), as_u32(qs), as_u32(qe)
        nq = len(qc)
        if out is None:
            offsets = np.empty(nq + 1, dtype=np.uint64)
            p, n = C.c_void_p(), C.c_uint64()
            check(lib.gtars_tokenize(self._h, ptr(qc), ptr(qs), ptr(qe), nq, ptr(offsets), C.byref(p), C.byref(n)))
            return offsets, take_u32(p, n.value)
        offsets, ids = out
        if offsets.dtype != np.uint64 or ids.dtype != np.uint32 or len(offsets) < nq + 1 or not (
                offsets.flags.c_contiguous and ids.flags.c_contiguous):
            raise ValueError("out must be (uint64[>= nq+1], uint32[...]) C-contiguous arrays")
        n = C.c_uint64()
        rc = lib.gtars_tokenize_into(self._h, ptr(qc), ptr(qs), ptr(qe), nq, ptr(offsets), ptr(ids), len(ids), C.byref(n))
        if rc == _lib.ERR_CAPACITY:
            ids = np.empty(int(n.value), dtype=np.uint32)
            rc = lib.gtars_tokenize_into(self._h, ptr(qc), ptr(qs), ptr(qe), nq, ptr(offsets), ptr(ids), len(ids), C.byref(n))
        check(rc)
        return offsets[: nq + 1], ids[: int(n.value)]

    def count_overlaps(self, qc, qs, qe, min_overlap: Optional[int] = None) -> np.ndarray:
        qc, qs, qe = as_u32(qc), as_u32(qs), as_u32(qe)
        out = np.zeros(len(qc), dtype=np.uint32)
        hm, mo = _minargs(min_overlap)
        check(lib.gtars_count_overlaps(self._h, ptr(qc), ptr(qs), ptr(qe), len(qc), hm, mo, ptr(out)))
        return out

    def bits_count(self, qc, qs, qe) -> np.ndarray:
        """Bits::count (bits.rs:337-344) per query: u64, the reference's wrapping arithmetic included."""
        qc, qs, qe = as_u32(qc), as_u32(qs), as_u32(qe)
        out = np.zeros(len(qc), dtype=np.uint64)
        check(lib.gtars_bits_count(self._h, ptr(qc), ptr(qs), ptr(qe), len(qc), ptr(out)))
        return out

    def any_overlaps(self, qc, qs, qe, min_overlap: Optional[int] = None) -> np.ndarray:
        qc, qs, qe = as_u32(qc), as_u32(qs), as_u32(qe)
        out = np.zeros(len(qc), dtype=np.uint8)
        hm, mo = _minargs(min_overlap)
        check(lib.gtars_any_overlaps(self._h, ptr(qc), ptr(qs), ptr(qe), len(qc), hm, mo, ptr(out)))
        return out.astype(bool)

    def find_overlaps(self, qc, qs, qe, min_overlap: Optional[int] = None):
        """-> (offsets, starts, ends, vals) (find_overlaps_regions, multi_chrom_overlapper.rs:525-550)."""
        qc, qs, qe = as_u32(qc), as_u32(qs), as_u32(qe)
        nq = len(qc)
        offsets = np.zeros(nq + 1, dtype=np.uint64)
        ps, pe, pv, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
        hm, mo = _minargs(min_overlap)
        check(lib.gtars_find_overlaps(self._h, ptr(qc), ptr(qs), ptr(qe), nq, hm, mo, ptr(offsets), C.byref(ps),
                                      C.byref(pe), C.byref(pv), C.byref(n)))
        return offsets, take_u32(ps, n.value), take_u32(pe, n.value), take_u32(pv, n.value)

    def find_overlap_indices(self, qc, qs, qe, min_overlap: Optional[int] = None):
        """IndexedRegionSet::find_overlaps (indexed_region_set.rs:246-263) -> (offsets, sorted unique idx)."""
        qc, qs, qe = as_u32(qc), as_u32(qs), as_u32(qe)
        nq = len(qc)
        offsets = np.zeros(nq + 1, dtype=np.uint64)
        p, n = C.c_void_p(), C.c_uint64()
        hm, mo = _minargs(min_overlap)
        check(lib.gtars_find_overlap_indices(self._h, ptr(qc), ptr(qs), ptr(qe), nq, hm, mo, ptr(offsets),
                                             C.byref(p), C.byref(n)))
        return offsets, take_u32(p, n.value)

    def subset_by_overlaps(self, qc, qs, qe, min_overlap: Optional[int] = None):
        """MultiChromOverlapper::subset_by_overlaps (multi_chrom_overlapper.rs:454-478): the index's intervals hit by any query
        -> (chrom ids, starts, ends), de-duplicated, sorted by (chrom id, start, end)."""
        qc, qs, qe = as_u32(qc), as_u32(qs), as_u32(qe)
        pc, ps, pe, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
        hm, mo = _minargs(min_overlap)
        check(lib.gtars_subset_by_overlaps(self._h, ptr(qc), ptr(qs), ptr(qe), len(qc), hm, mo, C.byref(pc), C.byref(ps),
                                           C.byref(pe), C.byref(n)))
        return take_u32(pc, n.value), take_u32(ps, n.value), take_u32(pe, n.value)

    def subset_source_indices(self, qc, qs, qe, min_overlap: Optional[int] = None) -> np.ndarray:
        """IndexedRegionSet::subset_by_overlaps / intersect_all (indexed_region_set.rs:201-230): the source rows (vals) hit
        by any query, ascending and unique."""
        qc, qs, qe = as_u32(qc), as_u32(qs), as_u32(qe)
        p, n = C.c_void_p(), C.c_uint64()
        hm, mo = _minargs(min_overlap)
        check(lib.gtars_subset_source_indices(self._h, ptr(qc), ptr(qs), ptr(qe), len(qc), hm, mo, C.byref(p), C.byref(n)))
        return take_u32(p, n.value)

    # -- device-pointer queries --------------------------------------------------
    def mark_overlapped_device(self, d_qc: int, d_qs: int, d_qe: int, nq: int, d_mark: int, min_overlap: Optional[int] = None,
                               stream: int = 0):
        """bit p of d_mark (ceil(len / 32) u32 words) = the interval at stored position p is hit by some query"""
        hm, mo = _minargs(min_overlap)
        check(lib.gtars_mark_overlapped_device(self._h, d_qc, d_qs, d_qe, nq, hm, mo, d_mark, stream))

    TOK_AUTO, TOK_NARROW, TOK_WIDE = 0, 1, 2  # gtars_amd.h: which build of the fused tokenizer a launch runs
    TOK_SORTED = 4  # OR-ed in: the batch is in (chromosome, start) order -> the sweep form of the tokenizer

    def tokenize_device(self, d_qc: int, d_qs: int, d_qe: int, nq: int, d_offsets: int, d_ids: int,
                        ids_capacity: int, stream: int = 0, sync: bool = True, hint: int = 0) -> Optional[int]:
        """Single fused pass on device buffers.  Returns H when ``sync`` (else None).  ``hint``: TOK_NARROW for batches of about
        one id per query whatever the id buffer's size, TOK_WIDE for hit-heavy ones; TOK_AUTO decides by the capacity; ``| TOK_SORTED``
        for a batch in (chromosome, start) order (a file-loaded RegionSet, a sorted BED file): the sweep form."""
        total = C.c_uint64()
        check(lib.gtars_tokenize_device_ex(self._h, d_qc, d_qs, d_qe, nq, d_offsets, d_ids, ids_capacity,
                                           C.byref(total) if sync else None, stream, hint))
        return int(total.value) if sync else None

    def fill_device(self, d_qc: int, d_qs: int, d_qe: int, nq: int, d_offsets: int, d_ids: int, stream: int = 0,
                    total_hits: Optional[int] = None):
        """ids for the offsets of a sizing pass; ``total_hits`` (what the caller read back to size ``d_ids``) bounds the writes
        and picks the tokenizer build"""
        if total_hits is None:
            check(lib.gtars_fill_device(self._h, d_qc, d_qs, d_qe, nq, d_offsets, d_ids, stream))
        else:
            check(lib.gtars_fill_device_n(self._h, d_qc, d_qs, d_qe, nq, d_offsets, d_ids, total_hits, stream))

    def count_overlaps_device(self, d_qc: int, d_qs: int, d_qe: int, nq: int, d_counts: int,
                              min_overlap: Optional[int] = None, stream: int = 0):
        hm, mo = _minargs(min_overlap)
        check(lib.gtars_count_overlaps_device(self._h, d_qc, d_qs, d_qe, nq, hm, mo, d_counts, stream))


class IgdIndex:
    """IGD database resident in HBM (gtars-igd/src/igd.rs), one record per stored interval."""

    def __init__(self, chrom, start, end, file_idx, value=None, n_chrom: Optional[int] = None,
                 n_files: Optional[int] = None):
        chrom = as_u32(chrom)
        start = np.ascontiguousarray(start, dtype=np.int32)
        end = np.ascontiguousarray(end, dtype=np.int32)
        file_idx = as_u32(file_idx)
        v = None if value is None else np.ascontiguousarray(value, dtype=np.int32)
        n = len(chrom)
        if n_chrom is None:
            n_chrom = int(chrom.max()) + 1 if n else 0
        if n_files is None:
            n_files = int(file_idx.max()) + 1 if n else 0
        h = C.c_void_p()
        check(lib.gtars_igd_build(ptr(chrom), ptr(start), ptr(end), ptr(v) if v is not None else None,
                                  ptr(file_idx), n, int(n_chrom), int(n_files), C.byref(h)))
        self._h = h
        self.n_files = int(n_files)
        self.n_chrom = int(n_chrom)

    def close(self):
        if getattr(self, "_h", None):
            lib.gtars_igd_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(lib.gtars_igd_len(self._h))

    def total_records(self, nbp: int = 16384) -> int:
        return int(lib.gtars_igd_total_records(self._h, nbp))

    def count_set_overlaps(self, qc, qs, qe, min_overlap: int = 1) -> np.ndarray:
        return self._count(qc, qs, qe, min_overlap, 0)

    def count_region_hits(self, qc, qs, qe, min_overlap: int = 1) -> np.ndarray:
        return self._count(qc, qs, qe, min_overlap, 1)

    def _count(self, qc, qs, qe, min_overlap, binary):
        qc, qs, qe = as_u32(qc), as_u32(qs), as_u32(qe)
        hits = np.zeros(self.n_files, dtype=np.uint64)
        check(lib.gtars_igd_count(self._h, ptr(qc), ptr(qs), ptr(qe), len(qc), int(min_overlap), binary, ptr(hits)))
        return hits

    def count_device(self, d_qc: int, d_qs: int, d_qe: int, nq: int, d_hits: int, min_overlap: int = 1,
                     binary: bool = False, stream: int = 0):
        check(lib.gtars_igd_count_device(self._h, d_qc, d_qs, d_qe, nq, int(min_overlap), int(binary), d_hits, stream))

    def count_sets(self, sets, min_overlap: int = 1, binary: bool = False) -> np.ndarray:
        """Several query sets [(chrom, start, end), ...] in one call -> u64[len(sets), n_files]; row k is what
        count_set_overlaps / count_region_hits returns for set k alone (up to 4 sets share one pass over the database:
        gtars_igd_count_sets, the count step of run_lola, gtars-lola/src/enrichment.rs:198-221)."""
        cols = [[as_u32(x[j]) for x in sets] for j in range(3)]
        off = np.zeros(len(sets) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(c) for c in cols[0]])
        qc, qs, qe = (np.ascontiguousarray(np.concatenate(c)) if len(sets) else np.zeros(0, dtype=np.uint32) for c in cols)
        hits = np.zeros((len(sets), self.n_files), dtype=np.uint64)
        if len(sets):
            check(lib.gtars_igd_count_sets(self._h, ptr(qc), ptr(qs), ptr(qe), ptr(off), len(sets), int(min_overlap), int(binary), ptr(hits)))
        return hits

    def count_sets_device(self, d_qc: int, d_qs: int, d_qe: int, set_off, d_hits: int, min_overlap: int = 1,
                          binary: bool = False, stream: int = 0):
        """Device form: the concatenated batch on the device, `set_off` a host sequence of len(sets) + 1 row offsets, d_hits
        u64[len(sets) * n_files].  Asynchronous on `stream`."""
        off = np.ascontiguousarray(set_off, dtype=np.uint64)
        check(lib.gtars_igd_count_sets_device(self._h, d_qc, d_qs, d_qe, ptr(off), len(off) - 1, int(min_overlap), int(binary), d_hits, stream))

    def count_overlaps_per_query(self, qc, qs, qe, min_overlap: int = 1) -> np.ndarray:
        qc, qs, qe = as_u32(qc), as_u32(qs), as_u32(qe)
        out = np.zeros(len(qc), dtype=np.uint32)
        check(lib.gtars_igd_count_per_query(self._h, ptr(qc), ptr(qs), ptr(qe), len(qc), int(min_overlap), ptr(out)))
        return out

    def find_overlaps_regionset(self, qc, qs, qe, min_overlap: int = 1):
        qc, qs, qe = as_u32(qc), as_u32(qs), as_u32(qe)
        pq, ps, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        check(lib.gtars_igd_find_pairs(self._h, ptr(qc), ptr(qs), ptr(qe), len(qc), int(min_overlap), C.byref(pq),
                                       C.byref(ps), C.byref(n)))
        return take_u32(pq, n.value), take_u32(ps, n.value)
