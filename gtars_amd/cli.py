"""Text front ends of the two CLI commands that sit directly on the hot path:

  python -m gtars_amd overlaprs --universe U.bed[.gz] --query Q.bed[.gz] [--backend bits|ailist]
      gtars-cli/src/overlaprs/handlers.rs:21-157: one line  chr<TAB>hit.start<TAB>hit.end  per hit, queries in file
      order, hits in the backend's find order (Bits: stored order; AIList: sub-list major, descending), queries on
      chromosomes the universe does not have are skipped.
  python -m gtars_amd igd create --output DIR --filelist DIR|list.txt|- [--dbname igd_database]
  python -m gtars_amd igd search --database DB.igd --query Q.bed[.gz]
      gtars-cli/src/igd/handlers.rs:11-98: the legacy TSV  index / number of regions / number of hits / File_name  for
      the files with hits, then  Total: N.

Same rules as the reference: fields are split on TAB only, coordinates must parse as u32 (``+5`` is accepted, blanks and
signs are not), every line counts (no header skipping in overlaprs).  The whole query file is ONE batch on the device.
"""
from __future__ import annotations

import argparse
import os
import sys
from typing import List, Sequence, TextIO, Tuple

import numpy as np

from ._lib import KIND_AILIST, KIND_BITS, UNKNOWN_CHROM


def read_bed3_lines(path: str) -> Tuple[List[str], np.ndarray, np.ndarray, np.ndarray]:
    """chr / start / end of EVERY line, in file order (handlers.rs:64-92, 123-139), through the C++ chunked reader
    (gtars_bed3_lines_read): -> (chromosome names in first-seen order, chromosome id per line, starts, ends)."""
    import ctypes as C

    from . import _lib

    h = C.c_void_p()
    if _lib.lib.gtars_bed3_lines_read(str(path).encode(), C.byref(h)) != 0:
        raise ValueError(_lib.last_error())
    try:
        n = int(_lib.lib.gtars_fragments_len(h))

        def col(fn):
            if not n:
                return np.zeros(0, dtype=np.uint32)
            return np.ctypeslib.as_array(C.cast(fn(h), C.POINTER(C.c_uint32)), shape=(n,)).copy()

        names = [_lib.lib.gtars_fragments_chrom_name(h, i).decode() for i in range(_lib.lib.gtars_fragments_n_chrom(h))]
        return names, col(_lib.lib.gtars_fragments_chrom_ids), col(_lib.lib.gtars_fragments_starts), col(_lib.lib.gtars_fragments_ends)
    finally:
        _lib.lib.gtars_fragments_free(h)


HIT_CHUNK = 1 << 20  # hit lines formatted per call of the C++ writer


def run_overlaprs(universe: str, query: str, backend: str = "bits", out: TextIO = sys.stdout) -> int:
    """-> number of hit lines written.  Text in and out go through the C++ host layer (chunked in-place parse, hit lines formatted
    and written a million at a time); the whole query file is one device batch in between -- so a malformed line anywhere in the
    query file fails the run BEFORE any hit line is written (the reference streams line by line, handlers.rs:114-155, and has
    written the hits of the lines in front of the bad one: INTEGRATION.md)."""
    import ctypes as C

    from . import _lib
    from .engine import OverlapIndex

    if backend not in ("bits", "ailist"):
        raise ValueError(f"Invalid backend type: {backend}. Valid options are 'bits' or 'ailist'")
    names, cid, us, ue = read_bed3_lines(universe)
    ix = OverlapIndex(cid, us, ue, None, n_chrom=len(names), kind=KIND_AILIST if backend == "ailist" else KIND_BITS)
    q_names, q_cid, qs, qe = read_bed3_lines(query)
    # the query file's chromosome ids -> the universe's (chromosomes the universe does not have are skipped)
    uid = {nme: i for i, nme in enumerate(names)}
    relabel = np.fromiter((uid.get(nme, UNKNOWN_CHROM) for nme in q_names), dtype=np.uint32, count=len(q_names))
    qc = relabel[q_cid] if len(q_cid) else np.zeros(0, dtype=np.uint32)
    offsets, hs, he, _ = ix.find_overlaps(qc, qs, qe)
    n = int(len(hs))
    if n:
        # formatted and written in chunks of HIT_CHUNK hits (one C++ buffer and one bytes object per chunk, straight to the
        # stream's binary layer when it has one): a 1e9-hit run needs tens of MB of text in flight, not tens of GB
        counts = np.diff(offsets.astype(np.int64))
        arr = (C.c_char_p * len(names))(*[nme.encode() for nme in names])
        hs, he = np.ascontiguousarray(hs), np.ascontiguousarray(he)
        sink = getattr(out, "buffer", None)
        if sink is not None:
            out.flush()
        q_lo = 0
        for lo in range(0, n, HIT_CHUNK):
            hi = min(n, lo + HIT_CHUNK)
            # the queries whose hits [lo, hi) belong to: offsets is their CSR
            q_hi = int(np.searchsorted(offsets, hi, side="left"))
            q_first = max(int(np.searchsorted(offsets, lo, side="right")) - 1, 0)
            rep = counts[q_first:q_hi].copy()
            rep[0] -= lo - int(offsets[q_first])
            rep[-1] -= int(offsets[q_hi]) - hi
            hit_chrom = np.ascontiguousarray(np.repeat(qc[q_first:q_hi], rep), dtype=np.uint32)
            text, ln = C.c_void_p(), C.c_uint64()
            _lib.check(_lib.lib.gtars_format_hit_lines(C.cast(arr, C.c_void_p), _lib.ptr(hit_chrom), _lib.ptr(hs[lo:hi]), _lib.ptr(he[lo:hi]),
                                                       hi - lo, C.byref(text), C.byref(ln)))
            try:
                data = C.string_at(text, ln.value)
            finally:
                _lib.lib.gtars_free(text)
            if sink is not None:
                sink.write(data)
            else:
                out.write(data.decode())
            q_lo = q_hi
        if sink is not None:
            sink.flush()
    return n


def resolve_bed_paths(filelist: str, stdin: TextIO = sys.stdin) -> List[str]:
    """handlers.rs:17-52: a .txt list, "-" / "stdin", or a directory of .bed / .gz files (sorted)."""
    if filelist.endswith(".txt"):
        with open(filelist) as fh:
            return [p.strip() for p in fh if p.strip()]
    if filelist in ("-", "stdin"):
        return [p.strip() for p in stdin if p.strip()]
    paths = [os.path.join(filelist, n) for n in os.listdir(filelist)
             if n.rsplit(".", 1)[-1] in ("bed", "gz") and "." in n and os.path.isfile(os.path.join(filelist, n))]
    return sorted(paths)


def run_igd_create(output: str, filelist: str, dbname: str = "igd_database", stdin: TextIO = sys.stdin) -> str:
    from .igd import Igd

    db = Igd.from_bed_files(resolve_bed_paths(filelist, stdin))
    path = os.path.join(output, f"{dbname}.igd")
    db.save(path)
    return path


def run_igd_search(database: str, query: str, out: TextIO = sys.stdout) -> int:
    """-> total number of hits."""
    from .igd import Igd
    from .models import RegionSet

    db = Igd.from_igd_file(database)
    hits = db.count_set_overlaps(RegionSet(query), 1)
    out.write("index\t number of regions\t number of hits\t File_name\n")
    for i, fi in enumerate(db.file_info):
        if int(hits[i]) > 0:
            out.write(f"{i}\t{fi.num_regions}\t{int(hits[i])}\t{fi.filename}\n")
    total = int(np.asarray(hits, dtype=np.uint64).sum())
    out.write(f"Total: {total}\n")
    return total


def main(argv: Sequence[str] = None) -> int:
    ap = argparse.ArgumentParser(prog="python -m gtars_amd", description=__doc__.split("\n\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    o = sub.add_parser("overlaprs", help="Find overlaps between a query file and a universe file")
    o.add_argument("--query", "-q", required=True)
    o.add_argument("--universe", "-u", required=True)
    o.add_argument("--backend", "-b", default="bits")
    g = sub.add_parser("igd", help="Create or search an integrated genome database (IGD)")
    gs = g.add_subparsers(dest="igd_cmd", required=True)
    c = gs.add_parser("create")
    c.add_argument("--output", required=True)
    c.add_argument("--filelist", required=True)
    c.add_argument("--dbname", default="igd_database")
    s = gs.add_parser("search")
    s.add_argument("--database", "-d", required=True)
    s.add_argument("--query", "-q", required=True)
    a = ap.parse_args(argv)
    if a.cmd == "overlaprs":
        run_overlaprs(a.universe, a.query, a.backend)
    elif a.igd_cmd == "create":
        run_igd_create(a.output, a.filelist, a.dbname)
    else:
        run_igd_search(a.database, a.query)
    return 0
