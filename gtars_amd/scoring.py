"""Fragment-file x consensus-peak count matrices on the GPU (gtars-scoring, SURVEY.md 8f "next" row).

``region_scoring_from_fragments`` mirrors gtars-scoring/src/fragment_scoring.rs:19-120 over a
``ConsensusSet`` (gtars-scoring/src/files.rs:60-131): the consensus BED is parsed + sorted like any
RegionSet, peak id = first-seen rank of the (chr,start,end,rest) region, one Bits index per chromosome.
ATAC mode probes the cut sites  [start+4, start+5)  and the INVERTED interval  [end-5, end-6)
(fragment_scoring.rs:58-84) -- the overlap test is applied to it unchanged, as in the reference
(Interval::overlap, gtars-core/src/models/interval.rs:47-50); ChIP mode probes the fragment itself.
All probes of all files go through ONE batched tokenization on the device; the scatter-add into the
matrix is a bincount over (file, peak) pairs.
"""
from __future__ import annotations

import glob as _glob
import gzip
from typing import Dict, List, Sequence, Union

import numpy as np

from ._lib import UNKNOWN_CHROM
from .engine import OverlapIndex
from .models import RegionSet

START_SHIFT = 4  # gtars-scoring/src/consts.rs
END_SHIFT = 5


class ConsensusSet:
    """gtars-scoring/src/files.rs:47-101"""

    def __init__(self, path: str):
        rs = RegionSet(path)
        regs = rs.regions
        ids: Dict[tuple, int] = {}
        vals = np.empty(len(regs), dtype=np.uint32)
        for i, r in enumerate(regs):  # generate_region_to_id_map: first-seen dense ids (gtars-core utils.rs:202-214)
            vals[i] = ids.setdefault((r.chr, r.start, r.end, r.rest), len(ids))
        self._len = len(regs)
        self.chrom_names = rs.chrom_names
        self._chrom_ids = {n: i for i, n in enumerate(self.chrom_names)}
        self.index = OverlapIndex(rs.chrom_ids, rs.starts, rs.ends, vals, n_chrom=len(self.chrom_names))

    def __len__(self) -> int:
        return self._len

    def chrom_id(self, name: str) -> int:
        return self._chrom_ids.get(name, UNKNOWN_CHROM)


def _read_fragments(path: str, cons: ConsensusSet):
    """Fragment::from_str (gtars-core/src/models/fragments.rs:16-41): whitespace split, u32 start/end/support."""
    opener = gzip.open if path.endswith(".gz") else open
    chrom, start, end, barcodes = [], [], [], []
    with opener(path, "rt") as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith("#"):
                continue
            p = line.split()
            s, e = int(p[1]), int(p[2])
            int(p[4])  # read_support must parse
            if not (0 <= s <= 0xFFFFFFFF and 0 <= e <= 0xFFFFFFFF):
                raise ValueError(f"bad fragment coordinates: {line}")
            chrom.append(cons.chrom_id(p[0]))
            start.append(s)
            end.append(e)
            barcodes.append(p[3])
    return (np.asarray(chrom, dtype=np.uint32), np.asarray(start, dtype=np.int64), np.asarray(end, dtype=np.int64), barcodes)


def _probes(start: np.ndarray, end: np.ndarray, mode: str):
    if mode == "atac":
        ns = (start + START_SHIFT) & 0xFFFFFFFF
        ne = (end - END_SHIFT) & 0xFFFFFFFF
        qs = np.stack([ns, ne], axis=1).reshape(-1)
        qe = np.stack([(ns + 1) & 0xFFFFFFFF, (ne - 1) & 0xFFFFFFFF], axis=1).reshape(-1)
        return qs.astype(np.uint32), qe.astype(np.uint32), 2
    if mode == "chip":
        return start.astype(np.uint32), end.astype(np.uint32), 1
    raise ValueError(f"Invalid scoring mode: {mode}")


def region_scoring_from_fragments(fragments: Union[str, Sequence[str]], consensus: Union[str, ConsensusSet],
                                  scoring_mode: str = "atac") -> np.ndarray:
    """-> u32 count matrix [n_files, n_peaks] (rows in sorted glob / given order)."""
    files = sorted(_glob.glob(fragments)) if isinstance(fragments, str) else list(fragments)
    cons = consensus if isinstance(consensus, ConsensusSet) else ConsensusSet(consensus)
    mode = scoring_mode.lower()
    if mode not in ("atac", "chip"):
        raise ValueError(f"Invalid scoring mode: {scoring_mode}")
    qc, qs, qe, row = [], [], [], []
    for i, path in enumerate(files):
        c, s, e, _ = _read_fragments(path, cons)
        ps, pe, k = _probes(s, e, mode)
        qc.append(np.repeat(c, k))
        qs.append(ps)
        qe.append(pe)
        row.append(np.full(len(ps), i, dtype=np.int64))
    mat = np.zeros((len(files), len(cons)), dtype=np.uint32)
    if not files or not sum(len(x) for x in qc):
        return mat
    qc, qs, qe, row = (np.concatenate(x) for x in (qc, qs, qe, row))
    offsets, ids = cons.index.tokenize(qc, qs, qe)
    per_query = np.diff(offsets.astype(np.int64))
    flat = np.repeat(row, per_query) * len(cons) + ids.astype(np.int64)
    mat += np.bincount(flat, minlength=mat.size).reshape(mat.shape).astype(np.uint32)
    return mat


def barcode_scoring_from_fragments(fragment_file: str, consensus: Union[str, ConsensusSet]) -> Dict[str, Dict[int, int]]:
    """fragment_scoring.rs:125-155: barcode -> {peak index -> count} (fragment itself as the probe)."""
    cons = consensus if isinstance(consensus, ConsensusSet) else ConsensusSet(consensus)
    c, s, e, barcodes = _read_fragments(fragment_file, cons)
    offsets, ids = cons.index.tokenize(c, s.astype(np.uint32), e.astype(np.uint32))
    out: Dict[str, Dict[int, int]] = {}
    for i, bc in enumerate(barcodes):
        lo, hi = int(offsets[i]), int(offsets[i + 1])
        if hi > lo:
            d = out.setdefault(bc, {})
            for v in ids[lo:hi]:
                d[int(v)] = d.get(int(v), 0) + 1
    return out
