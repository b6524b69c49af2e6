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


BAND_CELLS = 1 << 26  # cells of the device-resident band of the barcode x peak matrix (256 MB of u32)


def barcode_scoring_from_fragments(fragment_file: str, consensus: Union[str, ConsensusSet]) -> Dict[str, Dict[int, int]]:
    """fragment_scoring.rs:125-155: barcode -> {peak index -> count} (the fragment itself as the probe).

    On the device: the fragments are tokenized in one batch (``gtars_tokenize_device``), and every (barcode, peak) hit is
    scatter-added into a device-resident BAND of the barcode x peak matrix (``gtars_histogram_rows_device``: as many barcodes at a
    time as fit ``BAND_CELLS`` cells); a band comes back to the host once, and only its non-zero cells become dictionary entries."""
    import torch

    cons = consensus if isinstance(consensus, ConsensusSet) else ConsensusSet(consensus)
    c, s, e, b, barcodes = _read_fragments(fragment_file, cons)
    out: Dict[str, Dict[int, int]] = {}
    n_peaks, nq = len(cons), len(c)
    if nq == 0 or n_peaks == 0 or not barcodes:
        return out
    dev = torch.device("cuda", torch.cuda.current_device())
    stream = torch.cuda.current_stream().cuda_stream
    d = [torch.from_numpy(np.ascontiguousarray(x).view(np.int32)).to(dev) for x in (c, s, e, b)]
    offsets = torch.empty(nq + 1, dtype=torch.int64, device=dev)
    ids = torch.empty(2 * nq + 1024, dtype=torch.int32, device=dev)
    try:
        h = cons.index.tokenize_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, offsets.data_ptr(), ids.data_ptr(),
                                       ids.numel(), stream, sync=True)
    except _lib.CapacityError as err:  # more than two peaks per fragment on average: once more with the exact size
        ids = torch.empty(err.needed, dtype=torch.int32, device=dev)
        h = cons.index.tokenize_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, offsets.data_ptr(), ids.data_ptr(),
                                       ids.numel(), stream, sync=True)
    if h == 0:
        return out
    rows = max(1, min(len(barcodes), BAND_CELLS // n_peaks))
    band = torch.empty((rows, n_peaks), dtype=torch.int32, device=dev)
    for row0 in range(0, len(barcodes), rows):
        nr = min(rows, len(barcodes) - row0)
        band.zero_()
        check(lib.gtars_histogram_rows_device(offsets.data_ptr(), ids.data_ptr(), d[3].data_ptr(), nq, row0, nr, n_peaks, band.data_ptr(),
                                              stream))
        m = band[:nr].cpu().numpy().view(np.uint32)
        r, k = np.nonzero(m)
        for ri, ki, v in zip(r.tolist(), k.tolist(), m[r, k].tolist()):
            out.setdefault(barcodes[row0 + ri], {})[ki] = v
    return out


def write_sparse_counts_to_mtx(barcode_counts: Dict[str, Dict[int, int]], num_peaks: int, output_prefix: str) -> None:
    """write_sparse_counts_to_mtx (gtars-scoring/src/matrix_market.rs:26-92): ``{prefix}_matrix.mtx.gz`` (Matrix Market coordinate
    integer general; barcodes sorted, triplets sorted by (row, col), 1-based), ``{prefix}_barcodes.tsv.gz`` (one barcode per line,
    sorted) and ``{prefix}_features.tsv.gz`` (``peak_<i>`` per line) -- gzip level 6 like flate2's default."""
    import gzip

    barcodes = sorted(barcode_counts, key=lambda x: x.encode())  # Rust's String order is byte order
    trip = [(ri, col, cnt) for ri, bc in enumerate(barcodes) for col, cnt in barcode_counts[bc].items()]
    trip.sort(key=lambda t: (t[0], t[1]))
    with gzip.open(f"{output_prefix}_matrix.mtx.gz", "wt", compresslevel=6, newline="\n") as fh:
        fh.write("%%MatrixMarket matrix coordinate integer general\n")
        fh.write(f"{len(barcodes)} {num_peaks} {len(trip)}\n")
        fh.write("".join(f"{r + 1} {c + 1} {v}\n" for r, c, v in trip))
    with gzip.open(f"{output_prefix}_barcodes.tsv.gz", "wt", compresslevel=6, newline="\n") as fh:
        fh.write("".join(f"{bc}\n" for bc in barcodes))
    with gzip.open(f"{output_prefix}_features.tsv.gz", "wt", compresslevel=6, newline="\n") as fh:
        fh.write("".join(f"peak_{i}\n" for i in range(num_peaks)))


def write_count_matrix(matrix: np.ndarray, filename: str) -> None:
    """CountMatrix::write_to_file (gtars-scoring/src/counts.rs:89-105): one line per row, values joined by ",", gzip."""
    import gzip

    m = np.asarray(matrix)
    with gzip.open(filename, "wt", compresslevel=6, newline="\n") as fh:
        for row in m.reshape(m.shape[0], -1) if m.ndim else m.reshape(1, 1):
            fh.write(",".join(str(int(v)) for v in row) + "\n")
