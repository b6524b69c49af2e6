"""Fragment-file x consensus-peak count matrices on the GPU (gtars-scoring, SURVEY.md 8f "next" row).

``region_scoring_from_fragments`` mirrors gtars-scoring/src/fragment_scoring.rs:19-120 over a
``ConsensusSet`` (gtars-scoring/src/files.rs:60-131): the consensus BED is parsed + sorted like any
RegionSet, peak id = first-seen rank of the (chr,start,end,rest) region, one Bits index per chromosome.
ATAC mode probes the cut sites  [start+4, start+5)  and the INVERTED interval  [end-5, end-6)
(fragment_scoring.rs:58-84) -- the overlap test is applied to it unchanged, as in the reference
(Interval::overlap, gtars-core/src/models/interval.rs:47-50); ChIP mode probes the fragment itself.
Fragment files are read by the C++ in-place parser; every file's probes go through one batched tokenization on
the device and their token ids are scatter-added into the file's row of a device-resident matrix.
"""
from __future__ import annotations

import ctypes as C
import glob as _glob
from typing import Dict, Sequence, Union

import numpy as np

from . import _lib
from ._lib import UNKNOWN_CHROM, check, lib
from .engine import OverlapIndex
from .models import RegionSet

START_SHIFT = 4  # gtars-scoring/src/consts.rs
END_SHIFT = 5


class ConsensusSet:
    """gtars-scoring/src/files.rs:47-101"""

    def __init__(self, path: str):
        rs = RegionSet(path)
        # generate_region_to_id_map: first-seen dense ids over (chr, start, end, rest) (gtars-core utils.rs:202-214), in C++
        p = C.c_void_p()
        check(lib.gtars_regionset_dense_ids(rs._h, C.byref(p), None))
        vals = _lib.take_u32(p, len(rs))
        self._len = len(rs)
        self.chrom_names = rs.chrom_names
        self._chrom_ids = {n: i for i, n in enumerate(self.chrom_names)}
        self.index = OverlapIndex(rs.chrom_ids, rs.starts, rs.ends, vals, n_chrom=len(self.chrom_names))

    def __len__(self) -> int:
        return self._len

    def chrom_id(self, name: str) -> int:
        return self._chrom_ids.get(name, UNKNOWN_CHROM)


def _read_fragments(path: str, cons: ConsensusSet):
    """Fragment::from_str over a whole file (gtars-core/src/models/fragments.rs:16-41: whitespace split, u32 start / end /
    read support, '#' lines skipped) through the C++ in-place parser; chromosome ids translated to the consensus set's.
    -> (chrom u32, start u32, end u32, barcode ids u32, barcode names)."""
    h = C.c_void_p()
    if lib.gtars_fragments_read_strict(str(path).encode(), C.byref(h)) != 0:
        raise ValueError(_lib.last_error())
    try:
        n = int(lib.gtars_fragments_len(h))

        def col(fn):
            p = fn(h)
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n,)).copy() if n else np.zeros(0, np.uint32)

        fc, fs, fe, fb = (col(f) for f in (lib.gtars_fragments_chrom_ids, lib.gtars_fragments_starts, lib.gtars_fragments_ends,
                                           lib.gtars_fragments_barcode_ids))
        names = [lib.gtars_fragments_chrom_name(h, i).decode() for i in range(int(lib.gtars_fragments_n_chrom(h)))]
        barcodes = [lib.gtars_fragments_barcode_name(h, i).decode() for i in range(int(lib.gtars_fragments_n_barcodes(h)))]
    finally:
        lib.gtars_fragments_free(h)
    lut = np.asarray([cons.chrom_id(nm) for nm in names] or [UNKNOWN_CHROM], dtype=np.uint32)
    return (lut[fc] if n else fc), fs, fe, fb, barcodes


def _probes(start: np.ndarray, end: np.ndarray, mode: str):
    """u32 wrapping arithmetic of the reference (fragment_scoring.rs:58-84); returns probes per fragment too."""
    if mode == "atac":
        s64, e64 = start.astype(np.int64), end.astype(np.int64)
        ns = (s64 + START_SHIFT) & 0xFFFFFFFF
        ne = (e64 - END_SHIFT) & 0xFFFFFFFF
        qs = np.stack([ns, ne], axis=1).reshape(-1)
        qe = np.stack([(ns + 1) & 0xFFFFFFFF, (ne - 1) & 0xFFFFFFFF], axis=1).reshape(-1)
        return qs.astype(np.uint32), qe.astype(np.uint32), 2
    if mode == "chip":
        return start.astype(np.uint32), end.astype(np.uint32), 1
    raise ValueError(f"Invalid scoring mode: {mode}")


def region_scoring_from_fragments(fragments: Union[str, Sequence[str]], consensus: Union[str, ConsensusSet],
                                  scoring_mode: str = "atac") -> np.ndarray:
    """-> u32 count matrix [n_files, n_peaks] (rows in sorted glob / given order).

    Per file: the C++ parser yields the fragment columns, the probes are tokenized on the device
    (``gtars_tokenize_device``) and the token ids are scatter-added into the file's row of the DEVICE-resident matrix
    (``gtars_histogram_u32_device``); the matrix comes back to the host once, at the end."""
    import torch

    files = sorted(_glob.glob(fragments)) if isinstance(fragments, str) else list(fragments)
    cons = consensus if isinstance(consensus, ConsensusSet) else ConsensusSet(consensus)
    mode = scoring_mode.lower()
    if mode not in ("atac", "chip"):
        raise ValueError(f"Invalid scoring mode: {scoring_mode}")
    n_peaks = len(cons)
    if not files or n_peaks == 0:
        return np.zeros((len(files), n_peaks), dtype=np.uint32)
    dev = torch.device("cuda", torch.cuda.current_device())
    stream = torch.cuda.current_stream().cuda_stream
    mat = torch.zeros((len(files), n_peaks), dtype=torch.int32, device=dev)  # u32 counts, bit-identical
    for i, path in enumerate(files):
        c, s, e, _, _ = _read_fragments(path, cons)
        if len(c) == 0:
            continue
        ps, pe, k = _probes(s, e, mode)
        d = [torch.from_numpy(np.ascontiguousarray(x).view(np.int32)).to(dev) for x in (np.repeat(c, k), ps, pe)]
        nq = d[0].numel()
        offsets = torch.empty(nq + 1, dtype=torch.int64, device=dev)
        ids = torch.empty(2 * nq + 1024, dtype=torch.int32, device=dev)
        try:
            h = cons.index.tokenize_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, offsets.data_ptr(), ids.data_ptr(),
                                           ids.numel(), stream, sync=True)
        except _lib.CapacityError as err:  # more than two hits per probe on average: once more with the exact size
            ids = torch.empty(err.needed, dtype=torch.int32, device=dev)
            h = cons.index.tokenize_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, offsets.data_ptr(), ids.data_ptr(),
                                           ids.numel(), stream, sync=True)
        check(lib.gtars_histogram_u32_device(ids.data_ptr(), h, n_peaks, mat[i].data_ptr(), stream))
    return mat.cpu().numpy().view(np.uint32)


def barcode_scoring_from_fragments(fragment_file: str, consensus: Union[str, ConsensusSet]) -> Dict[str, Dict[int, int]]:
    """fragment_scoring.rs:125-155: barcode -> {peak index -> count} (fragment itself as the probe)."""
    cons = consensus if isinstance(consensus, ConsensusSet) else ConsensusSet(consensus)
    c, s, e, b, barcodes = _read_fragments(fragment_file, cons)
    offsets, ids = cons.index.tokenize(c, s, e)
    per_query = np.diff(offsets.astype(np.int64))
    # (barcode, peak) pairs of all hits, counted by one sort
    pair = np.repeat(b.astype(np.int64), per_query) * max(len(cons), 1) + ids.astype(np.int64)
    uniq, cnt = np.unique(pair, return_counts=True)
    out: Dict[str, Dict[int, int]] = {}
    n_peaks = max(len(cons), 1)
    for u, k in zip(uniq.tolist(), cnt.tolist()):
        out.setdefault(barcodes[u // n_peaks], {})[u % n_peaks] = k
    return out
