"""gtars-fragsplit: pseudobulking of scATAC fragment files by a barcode -> cluster map, and the
"fragsplit -> tokenizer" pipeline of BASELINE config 5.

Mirrors ``gtars_fragsplit::map::BarcodeToClusterMap`` (gtars-fragsplit/src/map.rs:8-81) and
``gtars_fragsplit::split::pseudobulk_fragment_files`` (split.rs:36-151) over the C ABI of ``include/gtars_amd_host.h``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Set

import numpy as np

from . import _lib
from ._lib import lib


def _check(st: int):
    if st != 0:
        raise RuntimeError(_lib.last_error())


class BarcodeToClusterMap:
    """``<file stem>+<barcode>`` -> cluster label, read from a two-column whitespace-separated file."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_file(cls, path: str) -> "BarcodeToClusterMap":
        h = C.c_void_p()
        _check(lib.gtars_barcode_map_from_file(os.fspath(path).encode(), C.byref(h)))
        return cls(h)

    def close(self):
        if getattr(self, "_h", None):
            lib.gtars_barcode_map_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        return int(lib.gtars_barcode_map_len(self._h))

    def n_clusters(self) -> int:
        return int(lib.gtars_barcode_map_n_clusters(self._h))

    def cluster_labels(self) -> List[str]:
        """labels in byte order (the order of the results of ``fragsplit_tokenize``)"""
        return [lib.gtars_barcode_map_cluster_label(self._h, i).decode() for i in range(self.n_clusters())]

    def get_cluster_labels(self) -> Set[str]:
        return set(self.cluster_labels())

    def get_cluster_from_barcode(self, barcode: str) -> Optional[str]:
        r = lib.gtars_barcode_map_lookup(self._h, barcode.encode())
        return r.decode() if r is not None else None


def pseudobulk_fragment_files(files: str, mapping: BarcodeToClusterMap, output: str) -> Dict[str, int]:
    """Split every fragment file of the folder ``files`` into ``output``/cluster_<id>.bed.gz according to ``mapping``.
    Returns {"reads": lines read, "written": lines routed to a cluster}."""
    n_reads, n_written = C.c_uint64(), C.c_uint64()
    _check(lib.gtars_fragsplit(os.fspath(files).encode(), mapping._h, os.fspath(output).encode(), C.byref(n_reads),
                               C.byref(n_written)))
    return {"reads": int(n_reads.value), "written": int(n_written.value)}


def _barcodes_of(ft_ptr, nb: int) -> List[str]:
    """the barcodes of one result in ONE call (a conversion per C string was a fifth of a 48-file call)"""
    if not nb:
        return []
    buf, n = C.c_void_p(), C.c_uint64()
    _check(lib.gtars_fragment_tokens_barcodes_joined(ft_ptr, C.byref(buf), C.byref(n)))
    try:
        return C.string_at(buf, n.value).decode().split("\n")
    finally:
        lib.gtars_free(buf)


class _ResultOwner:
    """keeps one cluster's C result alive for as long as a numpy array still looks at its memory (``as_arrays`` results are
    views, not copies: the copy of 15 MB of ids was a visible share of a 48-file call)"""

    def __init__(self, ft_ptr):
        self._addr = C.addressof(ft_ptr.contents)  # (the address itself: ft_ptr lives in the result array, which is freed at once)

    def __del__(self):
        if self._addr:
            lib.gtars_fragment_tokens_free(C.cast(C.c_void_p(self._addr), C.POINTER(_lib.FragmentTokens)))
            self._addr = 0


def _view(owner, ptr, n: int, ctype, dtype):
    if not n:
        return np.zeros(0, dtype=dtype)
    buf = (ctype * n).from_address(C.addressof(ptr.contents))
    buf._owner = owner  # (the array's base chain ends here)
    return np.frombuffer(buf, dtype=dtype)


def _collect_cluster_results(out, mapping: BarcodeToClusterMap, as_arrays: bool):
    labels = mapping.cluster_labels()
    res = {}
    owned = [False] * len(labels)
    try:
        for c, label in enumerate(labels):
            ft = out[c].contents
            nb = int(ft.n_barcodes)
            names = _barcodes_of(out[c], nb)
            if as_arrays:
                owner = _ResultOwner(out[c])
                owned[c] = True
                offs = _view(owner, ft.offsets, nb + 1, C.c_uint64, np.uint64)
                ids = _view(owner, ft.ids, int(offs[nb]), C.c_uint32, np.uint32)
                res[label] = (names, offs, ids)
            else:
                offs = np.ctypeslib.as_array(ft.offsets, shape=(nb + 1,))
                ids = np.ctypeslib.as_array(ft.ids, shape=(max(int(offs[nb]), 1),))
                res[label] = {names[b]: [int(v) for v in ids[int(offs[b]):int(offs[b + 1])]] for b in range(nb)}
    finally:
        for c in range(len(labels)):
            if not owned[c]:
                lib.gtars_fragment_tokens_free(out[c])
        lib.gtars_free(C.cast(out, C.c_void_p))
    return res


def fragsplit_tokenize(files: str, mapping: BarcodeToClusterMap, tokenizer, as_arrays: bool = False):
    """The fragsplit -> tokenizer pipeline without the intermediate files: {cluster label: {barcode: [ids]}}, for every
    cluster exactly what ``tokenize_fragment_file(output/cluster_<id>.bed.gz, tokenizer)`` would return.
    ``as_arrays``: {cluster: (barcodes, offsets uint64[nb+1], ids uint32[...])} instead (no per-id Python objects)."""
    out = C.POINTER(C.POINTER(_lib.FragmentTokens))()
    n_reads = C.c_uint64()
    _check(lib.gtars_fragsplit_tokenize(tokenizer._h, os.fspath(files).encode(), mapping._h, C.byref(out), C.byref(n_reads)))
    return _collect_cluster_results(out, mapping, as_arrays)


def list_fragment_files(files: str) -> List[str]:
    """The regular files of the folder ``files`` in byte order of their names: the order ``fragsplit_tokenize`` /
    ``pseudobulk_fragment_files`` visit them in (the reference's read_dir order is unspecified, split.rs:41-55)."""
    d = os.fspath(files)
    names = sorted((n for n in os.listdir(d) if os.path.isfile(os.path.join(d, n))), key=os.fsencode)
    return [os.path.join(d, n) for n in names]


def fragsplit_tokenize_files(paths: Sequence[str], mapping: BarcodeToClusterMap, tokenizer, as_arrays: bool = False):
    """``fragsplit_tokenize`` over an explicit list of fragment files, visited in the order given (one rank's run of the
    folder's sorted list: ``sharding.fragsplit_tokenize_sharded``)."""
    out = C.POINTER(C.POINTER(_lib.FragmentTokens))()
    n_reads = C.c_uint64()
    arr = (C.c_char_p * max(len(paths), 1))(*[os.fspath(p).encode() for p in paths])
    _check(lib.gtars_fragsplit_tokenize_files(tokenizer._h, arr, len(paths), mapping._h, C.byref(out), C.byref(n_reads)))
    return _collect_cluster_results(out, mapping, as_arrays)
