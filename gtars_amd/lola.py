"""``gtars.lola`` mirror: ``RegionDB`` and ``run_lola``.

The counting step -- the hot part of LOLA -- runs on the GPU:
``universe_hits`` / ``user_hits`` are ``Igd::count_region_hits`` support vectors
(gtars-igd/src/igd.rs:563-590, K5 binary kernel) and the 2x2 cells
``a, b, c, d`` (gtars-lola/src/enrichment.rs:214-220) come from the contingency
kernel.  The statistics tail stays on the host in f64 like the reference:
Fisher's exact p-value (enrichment.rs:19-53), CMLE odds ratio
(enrichment.rs:62-160, own Brent solver :400-486), min-ranks (:353-394), BH-FDR
(output.rs:35-113).

Parity note: the reference gets hypergeometric sf/cdf from the third-party
crate statrs 0.18, which is not part of the reference checkout; here they come
from scipy.stats.hypergeom, so ``pValueLog`` agrees to floating-point tolerance
only ("parity unpinned").  The integer columns (support, b, c, d)
are exact.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .igd import Igd
from .models import RegionSet



# --------------------------------------------------------------------------- RegionDB


class RegionDB:
    """gtars.lola.RegionDB (gtars-lola/src/database.rs:38-49)."""

    def __init__(self, igd: Igd, region_sets: List[RegionSet], region_anno: List[dict],
                 collection_anno: Optional[List[dict]] = None, db_location: Optional[str] = None):
        self.igd = igd
        self.region_sets = region_sets
        self._region_anno = region_anno
        self._collection_anno = collection_anno or []
        self.db_location = db_location

    @staticmethod
    def _anno(filename, collection=None, description=None, **kw) -> dict:
        d = {"filename": filename, "cellType": None, "description": description, "tissue": None, "dataSource": None,
             "antibody": None, "treatment": None, "collection": collection}
        d.update(kw)
        return d

    @staticmethod
    def from_bed_files(bed_files: Sequence[str], filenames: Optional[Sequence[str]] = None) -> "RegionDB":
        """gtars-python/src/lola/mod.rs:49-99"""
        names = list(filenames) if filenames is not None else [os.path.basename(p) or p for p in bed_files]
        sets, anno = [], []
        for i, p in enumerate(bed_files):
            try:
                rs = RegionSet(p)
            except RuntimeError as e:
                raise RuntimeError(f"Failed to read {p}: {e}") from None
            name = names[i] if i < len(names) else ""
            sets.append(rs)
            anno.append(RegionDB._anno(name))
        igd = Igd.from_named_region_sets([(a["filename"], rs) for a, rs in zip(anno, sets)])
        return RegionDB(igd, sets, anno)

    @staticmethod
    def from_folder(db_path: str, collections: Optional[Sequence[str]] = None, limit: Optional[int] = None) -> "RegionDB":
        """RegionDB::from_lola_folder (database.rs:52-181)."""
        if not os.path.isdir(db_path):
            raise RuntimeError(f"Failed to load RegionDB: {db_path} is not a directory")
        colls = sorted(
            d for d in os.listdir(db_path)
            if os.path.isdir(os.path.join(db_path, d, "regions")) and (collections is None or d in collections)
        )
        sets, anno, canno = [], [], []
        for coll in colls:
            cpath = os.path.join(db_path, coll)
            canno.append(_parse_collection_txt(os.path.join(cpath, "collection.txt"), coll))
            index = {a["filename"]: a for a in _parse_index_txt(os.path.join(cpath, "index.txt"), coll)}
            rdir = os.path.join(cpath, "regions")
            files = sorted(f for f in os.listdir(rdir) if os.path.isfile(os.path.join(rdir, f)))
            loaded = 0
            for f in files:
                if limit is not None and loaded >= limit:
                    break
                try:
                    rs = RegionSet(os.path.join(rdir, f))
                except RuntimeError:
                    continue  # "Warning: skipping ..."
                a = dict(index.get(f) or RegionDB._anno(f, collection=coll))
                if a.get("description") is None:
                    a["description"] = coll
                sets.append(rs)
                anno.append(a)
                loaded += 1
        igd = Igd.from_named_region_sets([(a["filename"], rs) for a, rs in zip(anno, sets)])
        return RegionDB(igd, sets, anno, canno, db_path)

    @property
    def num_region_sets(self) -> int:
        return len(self.region_sets)

    def list_region_sets(self, collections: Optional[Sequence[str]] = None) -> List[str]:
        return [a["filename"] for a in self._region_anno if collections is None or a.get("collection") in collections]

    def get_region_sets(self, indices: Optional[Sequence[int]] = None) -> List[RegionSet]:
        idx = range(len(self.region_sets)) if indices is None else indices
        return [self.region_sets[i] for i in idx if 0 <= i < len(self.region_sets)]

    @property
    def region_anno(self) -> List[dict]:
        return [dict(a) for a in self._region_anno]

    @property
    def collection_anno(self) -> List[dict]:
        return [dict(a) for a in self._collection_anno]

    def __repr__(self) -> str:
        return f"RegionDB({self.num_region_sets} region sets, {self.igd.num_contigs()} contigs)"


def _read_tsv(path: str) -> Tuple[List[str], List[List[str]]]:
    if not os.path.exists(path):
        return [], []
    with open(path, "rt") as f:
        lines = [l.rstrip("\n").rstrip("\r") for l in f]
    lines = [l for l in lines if l.strip()]
    if not lines:
        return [], []
    return lines[0].split("\t"), [l.split("\t") for l in lines[1:]]


def _parse_collection_txt(path: str, name: str) -> dict:
    out = {"collectionname": name, "collector": "", "date": "", "source": "", "description": ""}
    hdr, rows = _read_tsv(path)
    if rows:
        for k, v in zip(hdr, rows[0]):
            k = k.strip().lower()
            if k in out and k != "collectionname":
                out[k] = v.strip()
    return out


def _parse_index_txt(path: str, coll: str) -> List[dict]:
    hdr, rows = _read_tsv(path)
    cols = {h.strip().lower(): i for i, h in enumerate(hdr)}
    if "filename" not in cols:
        return []
    key = {"celltype": "cellType", "description": "description", "tissue": "tissue", "datasource": "dataSource",
           "antibody": "antibody", "treatment": "treatment"}
    out = []
    for r in rows:
        def get(c):
            i = cols.get(c)
            v = r[i].strip() if i is not None and i < len(r) else ""
            return v or None
        fn = get("filename")
        if not fn:
            continue
        a = RegionDB._anno(fn, collection=coll)
        for c, k in key.items():
            a[k] = get(c)
        out.append(a)
    return out


# --------------------------------------------------------------------------- statistics (host f64, compiled: csrc/lola_stats.cpp)


def fisher_pvalue(a: int, b: int, c: int, d: int, enrichment: bool = True) -> float:
    """ContingencyTable::fisher_pvalue (enrichment.rs:19-53)."""
    from ._lib import lib

    return float(lib.gtars_lola_fisher_pvalue(int(a), int(b), int(c), int(d), 0 if enrichment else 1))


def p_value_log(a: int, b: int, c: int, d: int, enrichment: bool = True) -> float:
    """ContingencyTable::p_value_log (enrichment.rs:166-169): -log10(p + 1e-322)."""
    return -math.log10(fisher_pvalue(a, b, c, d, enrichment) + 1e-322)


def odds_ratio(a: int, b: int, c: int, d: int) -> float:
    """ContingencyTable::odds_ratio (enrichment.rs:62-160): the conditional maximum-likelihood estimate of the odds ratio,
    as R's fisher.test reports it -- the omega for which the noncentral hypergeometric distribution of the table's margins
    has mean a.  Same definition and the same edge values (NaN for a one-point support, 0 / inf at the ends) as the
    reference; the numerics are the library's own (csrc/lola_stats.cpp): Newton in theta = log(omega) inside a sign-change
    bracket, mean and variance summed over the window of terms that matter by the exact term ratio.  (The reference finds
    the root of the same equation in omega with Brent's method, to an absolute 1e-8 in omega; its own tests pin the value
    to 1e-3.)"""
    from ._lib import lib

    return float(lib.gtars_lola_odds_ratio(int(a), int(b), int(c), int(d)))


def _rank_results(rows: List[dict]) -> None:
    """rank_results (enrichment.rs:353-394) on one user set's rows (gtars_lola_rank)."""
    from ._lib import check, lib

    n = len(rows)
    if n == 0:
        return
    pv = np.array([r["pValueLog"] for r in rows], dtype=np.float64)
    orr = np.array([r["oddsRatio"] for r in rows], dtype=np.float64)
    sup = np.array([r["support"] for r in rows], dtype=np.uint64)
    rk = [np.empty(n, dtype=np.uint32) for _ in range(4)]
    mean = np.empty(n, dtype=np.float64)
    check(lib.gtars_lola_rank(pv.ctypes.data, orr.ctypes.data, sup.ctypes.data, n, *[x.ctypes.data for x in rk], mean.ctypes.data))
    for i, r in enumerate(rows):
        r["rnkPV"], r["rnkOR"], r["rnkSup"], r["maxRnk"] = int(rk[0][i]), int(rk[1][i]), int(rk[2][i]), int(rk[3][i])
        r["meanRnk"] = float(mean[i])


def _apply_fdr(rows: List[dict]) -> None:
    """apply_fdr_correction (output.rs:35-113): Benjamini-Hochberg per user set (gtars_lola_fdr)."""
    from ._lib import check, lib

    n = len(rows)
    if n == 0:
        return
    pv = np.array([r["pValueLog"] for r in rows], dtype=np.float64)
    us = np.array([r["userSet"] for r in rows], dtype=np.uint64)
    q = np.empty(n, dtype=np.float64)
    check(lib.gtars_lola_fdr(pv.ctypes.data, us.ctypes.data, n, q.ctypes.data))
    for i, r in enumerate(rows):
        r["qValue"] = float(q[i])


def lola_stats(a, b, c, d, enrichment: bool = True) -> Dict[str, np.ndarray]:
    """Everything run_lola derives from the cells, for all tables at once (gtars_lola_stats, threaded over the tables).
    a, b, c, d: int64 [n_user_sets, n_db].  Returns the per-table columns in that layout plus ``order`` (flat row indices,
    user set * n_db + db set, in the reference's output order) and ``qValue``."""
    from ._lib import check, lib

    a, b, c, d = (np.ascontiguousarray(np.atleast_2d(x), dtype=np.int64) for x in (a, b, c, d))
    n_sets, n_db = a.shape
    out = {"pValueLog": np.empty(a.shape, np.float64), "oddsRatio": np.empty(a.shape, np.float64),
           "rnkPV": np.empty(a.shape, np.uint32), "rnkOR": np.empty(a.shape, np.uint32), "rnkSup": np.empty(a.shape, np.uint32),
           "maxRnk": np.empty(a.shape, np.uint32), "meanRnk": np.empty(a.shape, np.float64),
           "order": np.empty(a.size, np.uint64), "qValue": np.empty(a.shape, np.float64)}
    check(lib.gtars_lola_stats(a.ctypes.data, b.ctypes.data, c.ctypes.data, d.ctypes.data, n_db, n_sets, 0 if enrichment else 1,
                               *[out[k].ctypes.data for k in ("pValueLog", "oddsRatio", "rnkPV", "rnkOR", "rnkSup", "maxRnk",
                                                               "meanRnk", "order", "qValue")]))
    return out


# --------------------------------------------------------------------------- run_lola


def _as_regions(x):
    """What Igd.count_region_hits takes: a RegionSet goes through as columns (no per-region Python objects: a 1e6-region
    universe is three numpy arrays), a {"chr", "start", "end"} dict becomes a RegionSet by its columns (arrays stay arrays),
    anything else -- an iterable of (chr, start, end) tuples or of objects with .chr / .start / .end -- a list of tuples."""
    if isinstance(x, RegionSet):
        return x
    if isinstance(x, dict):
        return RegionSet.from_vectors(list(x["chr"]), x["start"], x["end"])
    x = list(x)
    if x and not isinstance(x[0], (tuple, list)):  # Region objects
        return [(r.chr, int(r.start), int(r.end)) for r in x]
    return [(r[0], int(r[1]), int(r[2])) for r in x]


def lola_counts(user_sets, universe, region_db: RegionDB, min_overlap: int = 1):
    """The GPU part of run_lola (enrichment.rs:198-221): support vectors and a, b, c, d per (user set, db set).

    Returns (universe_hits u64[F], [user_hits u64[F] ...], [(a, b, c, d) int64[F] ...])."""
    import torch

    from ._lib import check, lib

    igd = region_db.igd
    n_db = igd.num_files()
    if n_db == 0:
        raise RuntimeError("LOLA error: EmptyDatabase")
    uni = _as_regions(universe)
    if len(uni) == 0:
        raise RuntimeError("LOLA error: EmptyUniverse")
    # the universe and the user sets in one call: up to four sets share one pass over the region DB (the reference walks it once
    # per set, enrichment.rs:198-215)
    user_regs = [_as_regions(us) for us in user_sets]
    support = igd.count_region_hits_sets([uni] + user_regs, min_overlap)
    universe_hits = support[0]
    dev = torch.device("cuda", torch.cuda.current_device())
    d_uni = torch.from_numpy(universe_hits.astype(np.int64)).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    all_hits, outs = [], []
    for k, regs in enumerate(user_regs):
        hits = support[1 + k]
        d_user = torch.from_numpy(hits.astype(np.int64)).to(dev)
        out = torch.empty((4, n_db), dtype=torch.int64, device=dev)
        check(lib.gtars_lola_contingency_device(d_user.data_ptr(), d_uni.data_ptr(), n_db, len(regs), len(uni),
                                                *[out[j].data_ptr() for j in range(4)], stream))
        all_hits.append(hits)
        outs.append(out)
    cells = [tuple(o.cpu().numpy()) for o in outs]  # (the copies synchronise)
    return universe_hits, all_hits, cells


def run_lola(user_sets, universe, region_db: RegionDB, min_overlap: int = 1, direction: str = "enrichment") -> Dict[str, list]:
    """py_run_lola (gtars-python/src/lola/mod.rs:180-271): column dict, rows ordered like the reference.  Counts on the GPU
    (lola_counts), then ONE call into the library's compiled statistics tail for all tables (lola_stats): values, ranks
    inside a user set, the global order (pValueLog descending, then meanRnk ascending: enrichment.rs:285-294) and the
    q-values; what is left here is laying out the columns."""
    if direction in ("depletion", "less"):
        enrichment = False
    elif direction in ("enrichment", "greater"):
        enrichment = True
    else:
        raise ValueError("direction must be 'enrichment' or 'depletion'")
    _, _, cells = lola_counts(user_sets, universe, region_db, min_overlap)
    igd = region_db.igd
    cols = ["userSet", "dbSet", "collection", "pValueLog", "oddsRatio", "support", "rnkPV", "rnkOR", "rnkSup", "maxRnk",
            "meanRnk", "b", "c", "d", "description", "cellType", "tissue", "antibody", "treatment", "dataSource",
            "filename", "qValue", "size"]
    if not cells:
        return {c: [] for c in cols}
    a, b, c, d = (np.stack([cell[j] for cell in cells]) for j in range(4))
    n_db = a.shape[1]
    st = lola_stats(a, b, c, d, enrichment)
    order = st["order"].astype(np.int64)
    db = order % n_db
    out: Dict[str, list] = {"userSet": (order // n_db).tolist(), "dbSet": db.tolist()}
    for k, src in (("pValueLog", st["pValueLog"]), ("oddsRatio", st["oddsRatio"]), ("support", a), ("rnkPV", st["rnkPV"]),
                   ("rnkOR", st["rnkOR"]), ("rnkSup", st["rnkSup"]), ("maxRnk", st["maxRnk"]), ("meanRnk", st["meanRnk"]),
                   ("b", b), ("c", c), ("d", d), ("qValue", st["qValue"])):
        out[k] = src.reshape(-1)[order].tolist()
    # per database set: the annotation columns (output.rs:13-29) and the file name (enrichment.rs:249-253)
    anno = region_db.region_anno
    per_db: Dict[str, list] = {k: [] for k in ("collection", "description", "cellType", "tissue", "antibody", "treatment",
                                                 "dataSource", "filename", "size")}
    for f in range(n_db):
        an = anno[f] if f < len(anno) else {}
        desc = an.get("description")
        per_db["collection"].append(an.get("collection"))
        per_db["description"].append(desc[:80] if desc is not None else None)
        for k in ("cellType", "tissue", "antibody", "treatment", "dataSource"):
            per_db[k].append(an.get(k))
        per_db["filename"].append(igd.file_info[f].filename if f < len(igd.file_info) else "")
        per_db["size"].append(len(region_db.region_sets[f]) if f < len(region_db.region_sets) else 0)
    dbl = out["dbSet"]
    for k, v in per_db.items():
        out[k] = [v[f] for f in dbl]
    return {c: out[c] for c in cols}


def _rust_exp6(x: float) -> str:
    """format!("{:.6e}", x): mantissa with six decimals, exponent without sign padding (1.000000e-5, 3.000000e0)."""
    m, e = f"{x:.6e}".split("e")
    return f"{m}e{int(e)}"


def write_results_tsv(out, results: Dict[str, list]) -> None:
    """write_results_tsv (gtars-lola/src/output.rs:191-244): R LOLA's writeCombinedEnrichment layout, 1-based set indices,
    pValueLog / oddsRatio with 4 decimals, meanRnk with 2, qValue as %.6e or NA.  ``out``: a path or a text file object."""
    own = isinstance(out, (str, os.PathLike))
    f = open(out, "w", newline="") if own else out
    try:
        f.write("userSet\tdbSet\tcollection\tpValueLog\toddsRatio\tsupport\trnkPV\trnkOR\trnkSup\tmaxRnk\tmeanRnk\tb\tc\td\t"
                "description\tcellType\ttissue\tantibody\ttreatment\tdataSource\tfilename\tqValue\tsize\n")
        n = len(results["userSet"])

        def txt(col, i):
            v = results[col][i]
            return "" if v is None else str(v)

        def f4(v):
            return "NaN" if math.isnan(v) else ("inf" if v == math.inf else ("-inf" if v == -math.inf else f"{v:.4f}"))

        for i in range(n):
            q = results["qValue"][i]
            f.write("\t".join([
                str(results["userSet"][i] + 1), str(results["dbSet"][i] + 1), txt("collection", i), f4(results["pValueLog"][i]),
                f4(results["oddsRatio"][i]), str(results["support"][i]), str(results["rnkPV"][i]), str(results["rnkOR"][i]),
                str(results["rnkSup"][i]), str(results["maxRnk"][i]), f"{results['meanRnk'][i]:.2f}", str(results["b"][i]),
                str(results["c"][i]), str(results["d"][i]), txt("description", i), txt("cellType", i), txt("tissue", i),
                txt("antibody", i), txt("treatment", i), txt("dataSource", i), txt("filename", i),
                "NA" if q is None else _rust_exp6(q), str(results["size"][i])]) + "\n")
    finally:
        if own:
            f.close()


# --------------------------------------------------------------------------- universe helpers (gtars-lola/src/universe.rs)


def _to_region_set(x) -> RegionSet:
    if isinstance(x, RegionSet):
        return x
    if isinstance(x, dict):
        return RegionSet.from_vectors(list(x["chr"]), list(x["start"]), list(x["end"]))
    x = list(x)
    return RegionSet.from_vectors([r[0] for r in x], [int(r[1]) for r in x], [int(r[2]) for r in x])


def check_universe(user_sets, universe) -> Dict[str, list]:
    """check_universe_appropriateness (universe.rs:39-105) as py_check_universe returns it (gtars-python/src/lola/mod.rs:
    279-320): per user set the number of regions, how many overlap the universe (count_overlaps_per_query against an IGD
    of the universe, on the GPU), the coverage, the many-to-many count, and the reference's warning texts."""
    uni = _to_region_set(universe)
    igd = Igd.from_single_region_set(uni)
    out = {"userSet": [], "totalRegions": [], "regionsInUniverse": [], "coverage": [], "manyToMany": [], "warnings": []}
    for us_idx, us in enumerate(user_sets):
        rs = _to_region_set(us)
        total = len(rs)
        counts = np.asarray(igd.count_overlaps_per_query(rs, 1), dtype=np.int64)
        in_u, m2m = int((counts > 0).sum()), int((counts > 1).sum())
        cov = in_u / total if total else 0.0
        if cov < 0.5:
            out["warnings"].append(f"User set {us_idx}: only {cov * 100.0:.1f}% of regions overlap the universe. "
                                   "Consider using a more appropriate universe.")
        elif cov < 0.9:
            out["warnings"].append(f"User set {us_idx}: {cov * 100.0:.1f}% of regions overlap the universe. "
                                   "Some regions may not be represented.")
        if m2m > 0:
            out["warnings"].append(f"User set {us_idx}: {m2m} regions overlap multiple universe regions (many-to-many). "
                                   "Consider using redefine_user_sets() to eliminate artifacts.")
        out["userSet"].append(us_idx)
        out["totalRegions"].append(total)
        out["regionsInUniverse"].append(in_u)
        out["coverage"].append(cov)
        out["manyToMany"].append(m2m)
    return out


def redefine_user_sets(user_sets, universe) -> List[List[Tuple[str, int, int]]]:
    """redefine_user_sets (universe.rs:107-139): every user set replaced by the universe regions it overlaps
    (find_overlaps_regionset on the GPU, de-duplicated, sorted by (chr, start))."""
    uni = _to_region_set(universe)
    igd = Igd.from_single_region_set(uni)
    names, ids, st, en = uni.chrom_names, uni.chrom_ids, uni.starts, uni.ends
    out = []
    for us in user_sets:
        pairs = igd.find_overlaps_regionset(_to_region_set(us), 1)
        subj = list(dict.fromkeys(s for _, s in pairs))  # first occurrence wins, like the HashSet + push
        regs = [(names[int(ids[i])], int(st[i]), int(en[i])) for i in subj]
        regs.sort(key=lambda r: (r[0], r[1]))  # stable, like sort_by
        out.append(regs)
    return out


def build_restricted_universe(user_sets) -> List[Tuple[str, int, int]]:
    """build_restricted_universe (universe.rs:141-152): all user regions concatenated, then RegionSet::disjoin
    (gtars-core/src/models/region_set.rs:1051-1090): cut at every boundary, keep the pieces some region covers, sorted
    by (chr, start).  Host code: set algebra is outside the GPU path."""
    by_chr: Dict[str, List[Tuple[int, int]]] = {}
    for us in user_sets:
        rs = _to_region_set(us)
        names, ids, st, en = rs.chrom_names, rs.chrom_ids, rs.starts, rs.ends
        for c in range(len(names)):
            m = ids == c
            if m.any():
                by_chr.setdefault(names[c], []).extend(zip(st[m].tolist(), en[m].tolist()))
    result: List[Tuple[str, int, int]] = []
    for chrom, iv in by_chr.items():
        a = np.asarray(iv, dtype=np.int64)
        bounds = np.unique(a.reshape(-1))
        if len(bounds) < 2:
            continue
        # piece [bounds[i], bounds[i+1]) is covered iff some interval has start <= bounds[i] and bounds[i+1] <= end
        ok = a[:, 0] <= a[:, 1]  # an inverted interval contains no piece
        cover = np.zeros(len(bounds) + 1, dtype=np.int64)
        np.add.at(cover, np.searchsorted(bounds, a[ok, 0]), 1)
        np.add.at(cover, np.searchsorted(bounds, a[ok, 1]), -1)
        depth = np.cumsum(cover)[: len(bounds) - 1]
        for i in np.nonzero(depth > 0)[0]:
            result.append((chrom, int(bounds[i]), int(bounds[i + 1])))
    result.sort(key=lambda r: (r[0], r[1]))
    return result
