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


# --------------------------------------------------------------------------- statistics (host f64)


def fisher_pvalue(a: int, b: int, c: int, d: int, enrichment: bool = True) -> float:
    """ContingencyTable::fisher_pvalue (enrichment.rs:19-53)."""
    from scipy.stats import hypergeom

    n_pop, k_success, n_draws = a + b + c + d, a + b, a + c
    if n_pop == 0 or k_success == 0 or n_draws == 0:
        return 1.0
    if k_success > n_pop or n_draws > n_pop:
        return 1.0
    if enrichment:
        return 1.0 if a == 0 else float(hypergeom.sf(a - 1, n_pop, k_success, n_draws))
    return float(hypergeom.cdf(a, n_pop, k_success, n_draws))


def p_value_log(a: int, b: int, c: int, d: int, enrichment: bool = True) -> float:
    """ContingencyTable::p_value_log (enrichment.rs:166-169): -log10(p + 1e-322)."""
    return -math.log10(fisher_pvalue(a, b, c, d, enrichment) + 1e-322)


def odds_ratio(a: int, b: int, c: int, d: int) -> float:
    """ContingencyTable::odds_ratio (enrichment.rs:62-160): the conditional maximum-likelihood estimate of the odds ratio,
    as R's fisher.test reports it -- the omega for which the noncentral hypergeometric distribution of the table's margins
    has mean a.  Same definition and the same edge values (NaN for a one-point support, 0 / inf at the ends) as the
    reference; the numerics are this module's own: the support's log-weights come from lgamma, the equation is solved in
    theta = log(omega), where the mean is strictly increasing with the variance as its derivative, by Newton steps kept
    inside a sign-change bracket.  (The reference finds the root of the same equation in omega with Brent's method, to an
    absolute 1e-8 in omega; its own tests pin the value to 1e-3.)"""
    from scipy.special import gammaln, logsumexp

    m, n, k, x = a + c, b + d, a + b, a
    lo = k - n if k > n else 0
    hi = min(k, m)
    if lo == hi:
        return float("nan")
    if x == lo:
        return 0.0
    if x == hi:
        return float("inf")
    ys = np.arange(lo, hi + 1, dtype=np.float64)
    # log of C(m, y) * C(n, k - y) up to a constant
    lw = -(gammaln(ys + 1) + gammaln(m - ys + 1) + gammaln(k - ys + 1) + gammaln(n - k + ys + 1))

    def moments(theta: float):
        lv = lw + theta * ys
        p = np.exp(lv - logsumexp(lv))
        mu = float(np.dot(p, ys))
        return mu, float(np.dot(p, (ys - mu) ** 2))

    target = float(x)
    mu0, _ = moments(0.0)
    if abs(mu0 - target) < 1e-12:
        return 1.0
    # bracket the root: the mean runs from lo to hi as theta goes from -inf to +inf, and lo < x < hi here
    step = 1.0
    if mu0 < target:
        t_lo, t_hi = 0.0, step
        while moments(t_hi)[0] < target:
            t_lo, t_hi, step = t_hi, t_hi + 2.0 * step, 2.0 * step
    else:
        t_lo, t_hi = -step, 0.0
        while moments(t_lo)[0] > target:
            t_lo, t_hi, step = t_lo - 2.0 * step, t_lo, 2.0 * step
    theta = 0.5 * (t_lo + t_hi)
    for _ in range(200):
        mu, var = moments(theta)
        if mu < target:
            t_lo = theta
        else:
            t_hi = theta
        nxt = theta - (mu - target) / var if var > 0.0 else float("nan")
        if not (t_lo < nxt < t_hi):  # Newton left the bracket (flat tail of the mean): bisect
            nxt = 0.5 * (t_lo + t_hi)
        if abs(nxt - theta) <= 1e-13 * max(1.0, abs(theta)) or t_hi - t_lo <= 1e-14 * max(1.0, abs(theta)):
            theta = nxt
            break
        theta = nxt
    return math.exp(theta)


def _min_ranks(order: List[int], key) -> Dict[int, int]:
    """assign_min_ranks_* (enrichment.rs:310-351): ties.method = "min" on a pre-sorted index list."""
    ranks: Dict[int, int] = {}
    rank = 1
    for pos, idx in enumerate(order):
        if pos > 0:
            p, c = key(order[pos - 1]), key(idx)
            tied = (p == c) or (isinstance(p, float) and isinstance(c, float) and math.isnan(p) and math.isnan(c))
            if not tied:
                rank = pos + 1
        ranks[idx] = rank
    return ranks


def _rank_results(rows: List[dict]) -> None:
    """rank_results (enrichment.rs:353-394); Python's sort is stable like Rust's sort_by."""
    n = len(rows)
    idx = list(range(n))
    by_pv = sorted(idx, key=lambda i: -rows[i]["pValueLog"])
    r_pv = _min_ranks(by_pv, lambda i: rows[i]["pValueLog"])

    def or_key(i):
        v = rows[i]["oddsRatio"]
        return (1, 0.0) if math.isnan(v) else (0, -v)

    by_or = sorted(idx, key=or_key)
    r_or = _min_ranks(by_or, lambda i: rows[i]["oddsRatio"])
    by_sup = sorted(idx, key=lambda i: -rows[i]["support"])
    r_sup = _min_ranks(by_sup, lambda i: rows[i]["support"])
    for i in idx:
        rows[i]["rnkPV"], rows[i]["rnkOR"], rows[i]["rnkSup"] = r_pv[i], r_or[i], r_sup[i]
        rows[i]["maxRnk"] = max(r_pv[i], r_or[i], r_sup[i])
        rows[i]["meanRnk"] = (r_pv[i] + r_or[i] + r_sup[i]) / 3.0


def _apply_fdr(rows: List[dict]) -> None:
    """apply_fdr_correction (output.rs:35-113): Benjamini-Hochberg per user set."""
    if not rows:
        return
    for us in range(max(r["userSet"] for r in rows) + 1):
        idx = [i for i, r in enumerate(rows) if r["userSet"] == us]
        if not idx:
            continue
        n = len(idx)
        idx.sort(key=lambda i: -rows[i]["pValueLog"])
        p = [0.0 if rows[i]["pValueLog"] == float("inf") else 10.0 ** (-rows[i]["pValueLog"]) for i in idx]
        q = [0.0] * n
        q[n - 1] = min(p[n - 1] * n / n, 1.0)
        for i in range(n - 2, -1, -1):
            q[i] = min(min(p[i] * n / (i + 1), q[i + 1]), 1.0)
        for j, i in enumerate(idx):
            rows[i]["qValue"] = q[j]


# --------------------------------------------------------------------------- run_lola


def _as_regions(x):
    """What Igd.count_region_hits takes: a RegionSet goes through as columns (no per-region Python objects: a 1e6-region
    universe is three numpy arrays), a {"chr", "start", "end"} dict becomes a RegionSet, anything else a list of tuples."""
    if isinstance(x, RegionSet):
        return x
    if isinstance(x, dict):
        return RegionSet.from_vectors(list(x["chr"]), list(x["start"]), list(x["end"]))
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
    all_hits, cells = [], []
    for k, regs in enumerate(user_regs):
        hits = support[1 + k]
        d_user = torch.from_numpy(hits.astype(np.int64)).to(dev)
        out = [torch.empty(n_db, dtype=torch.int64, device=dev) for _ in range(4)]
        check(lib.gtars_lola_contingency_device(d_user.data_ptr(), d_uni.data_ptr(), n_db, len(regs), len(uni),
                                                *[o.data_ptr() for o in out], stream))
        torch.cuda.synchronize()
        all_hits.append(hits)
        cells.append(tuple(o.cpu().numpy() for o in out))
    return universe_hits, all_hits, cells


def run_lola(user_sets, universe, region_db: RegionDB, min_overlap: int = 1, direction: str = "enrichment") -> Dict[str, list]:
    """py_run_lola (gtars-python/src/lola/mod.rs:180-271): column dict, rows ordered like the reference."""
    if direction in ("depletion", "less"):
        enrichment = False
    elif direction in ("enrichment", "greater"):
        enrichment = True
    else:
        raise ValueError("direction must be 'enrichment' or 'depletion'")
    _, _, cells = lola_counts(user_sets, universe, region_db, min_overlap)
    igd = region_db.igd
    rows_all: List[dict] = []
    for us_idx, (a, b, c, d) in enumerate(cells):
        rows = []
        for f in range(len(a)):
            av, bv, cv, dv = int(a[f]), int(b[f]), int(c[f]), int(d[f])
            if bv < 0 or cv < 0 or dv < 0:
                pv_log, orr = 0.0, float("nan")
            else:
                pv_log = p_value_log(av, bv, cv, dv, enrichment)
                orr = odds_ratio(av, bv, cv, dv)
            rows.append({"userSet": us_idx, "dbSet": f, "pValueLog": pv_log, "oddsRatio": orr, "support": av,
                         "b": bv, "c": cv, "d": dv, "qValue": None,
                         "filename": igd.file_info[f].filename if f < len(igd.file_info) else ""})
        _rank_results(rows)
        rows_all.extend(rows)
    # global order: pValueLog descending, then meanRnk ascending (enrichment.rs:285-294)
    rows_all.sort(key=lambda r: (-r["pValueLog"], r["meanRnk"]))
    anno = region_db.region_anno
    for r in rows_all:
        a = anno[r["dbSet"]] if r["dbSet"] < len(anno) else {}
        desc = a.get("description")
        r.update({"collection": a.get("collection"), "description": desc[:80] if desc is not None else None,
                  "cellType": a.get("cellType"), "tissue": a.get("tissue"), "antibody": a.get("antibody"),
                  "treatment": a.get("treatment"), "dataSource": a.get("dataSource"),
                  "size": len(region_db.region_sets[r["dbSet"]]) if r["dbSet"] < len(region_db.region_sets) else 0})
    _apply_fdr(rows_all)
    cols = ["userSet", "dbSet", "collection", "pValueLog", "oddsRatio", "support", "rnkPV", "rnkOR", "rnkSup", "maxRnk",
            "meanRnk", "b", "c", "d", "description", "cellType", "tissue", "antibody", "treatment", "dataSource",
            "filename", "qValue", "size"]
    return {c: [r.get(c) for r in rows_all] for c in cols}


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
