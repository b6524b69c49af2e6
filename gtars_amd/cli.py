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
import gzip
import os
import re
import sys
from typing import Dict, List, Sequence, TextIO, Tuple

import numpy as np

from ._lib import KIND_AILIST, KIND_BITS, UNKNOWN_CHROM

_U32 = re.compile(r"\+?[0-9]+\Z")


def _open_text(path: str):
    return gzip.open(path, "rt") if str(path).endswith(".gz") else open(path, "rt")


def _parse_u32(text: str, path: str, line_no: int) -> int:
    if not _U32.match(text) or int(text) > 0xFFFFFFFF:  # str::parse::<u32>
        raise ValueError(f"{path}:{line_no}: invalid digit found in string: {text!r}")
    return int(text)


def read_bed3_lines(path: str) -> Tuple[List[str], np.ndarray, np.ndarray]:
    """chr / start / end of EVERY line, in file order (handlers.rs:64-92, 123-139)."""
    chrs: List[str] = []
    starts: List[int] = []
    ends: List[int] = []
    with _open_text(path) as fh:
        for no, line in enumerate(fh, 1):
            line = line.rstrip("\n")
            if line.endswith("\r"):  # BufRead::lines strips "\r\n"
                line = line[:-1]
            f = line.split("\t")
            if len(f) < 3:
                raise ValueError(f"{path}:{no}: Missing {'start' if len(f) == 1 else 'end'} field")
            chrs.append(f[0])
            starts.append(_parse_u32(f[1], path, no))
            ends.append(_parse_u32(f[2], path, no))
    return chrs, np.asarray(starts, dtype=np.uint32), np.asarray(ends, dtype=np.uint32)


def run_overlaprs(universe: str, query: str, backend: str = "bits", out: TextIO = sys.stdout) -> int:
    """-> number of hit lines written."""
    from .engine import OverlapIndex

    if backend not in ("bits", "ailist"):
        raise ValueError(f"Invalid backend type: {backend}. Valid options are 'bits' or 'ailist'")
    uc, us, ue = read_bed3_lines(universe)
    names: Dict[str, int] = {}
    cid = np.fromiter((names.setdefault(c, len(names)) for c in uc), dtype=np.uint32, count=len(uc))
    ix = OverlapIndex(cid, us, ue, None, n_chrom=len(names), kind=KIND_AILIST if backend == "ailist" else KIND_BITS)
    qc_names, qs, qe = read_bed3_lines(query)
    qc = np.fromiter((names.get(c, UNKNOWN_CHROM) for c in qc_names), dtype=np.uint32, count=len(qc_names))
    offsets, hs, he, _ = ix.find_overlaps(qc, qs, qe)
    per_query = np.diff(offsets.astype(np.int64))
    if len(hs):
        chrom_of_hit = np.repeat(np.asarray(qc_names, dtype=object), per_query)
        out.write("".join(f"{c}\t{s}\t{e}\n" for c, s, e in zip(chrom_of_hit, hs.tolist(), he.tolist())))
    return int(len(hs))


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
