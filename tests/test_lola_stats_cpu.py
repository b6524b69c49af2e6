"""The statistics tail of LOLA (host f64): every known-answer / property test the reference holds for it, ported:
gtars-lola/src/output.rs:245-531 (BH-FDR, TSV writer, column pivot) and gtars-lola/src/enrichment.rs:490-710 (odds ratio,
Fisher p-values, ranking), plus an independent evaluation of the p-values (oracle.fisher_pvalue: pmf tail sums from
lgamma) against the product's (scipy.stats.hypergeom).  statrs 0.18 -- what the reference calls -- is absent, so parity
of pValueLog with the reference itself stays unpinned beyond these tests; the integer cells and everything derived from
them by exact rules (ranks, q-values given p-values) are pinned."""
import io
import math

import numpy as np
import pytest

import oracle
from gtars_amd import lola


def _rows(pvls, user_sets=None):
    user_sets = user_sets or [0] * len(pvls)
    return [{"userSet": u, "dbSet": i, "pValueLog": p, "oddsRatio": 1.0, "support": 10, "qValue": None}
            for i, (u, p) in enumerate(zip(user_sets, pvls))]


# ---- output.rs:271-305 test_fdr_basic
def test_fdr_basic():
    rows = _rows([3.0, 2.0, 1.0])
    lola._apply_fdr(rows)
    q = [r["qValue"] for r in rows]
    assert all(v is not None for v in q)
    for r in rows:
        assert r["qValue"] >= 10.0 ** (-r["pValueLog"]) - 1e-10
    assert q[0] <= q[1] + 1e-10 and q[1] <= q[2] + 1e-10
    # the exact BH values: p = (1e-3, 1e-2, 1e-1), n = 3 -> q = (min(3e-3, ...), min(1.5e-2, ...), 1e-1)
    assert q == pytest.approx([0.003, 0.015, 0.1], rel=1e-12)
    assert q == oracle.bh_qvalues([3.0, 2.0, 1.0])


# ---- output.rs:307-323 test_fdr_multiple_user_sets: corrected per user set
def test_fdr_multiple_user_sets():
    rows = _rows([5.0, 2.0, 3.0, 1.0], [0, 0, 1, 1])
    lola._apply_fdr(rows)
    assert all(r["qValue"] is not None for r in rows)
    assert [rows[0]["qValue"], rows[1]["qValue"]] == oracle.bh_qvalues([5.0, 2.0])
    assert [rows[2]["qValue"], rows[3]["qValue"]] == oracle.bh_qvalues([3.0, 1.0])
    assert rows[0]["qValue"] == pytest.approx(2e-5) and rows[3]["qValue"] == pytest.approx(0.1)


# ---- output.rs:325-333 test_fdr_single_result
def test_fdr_single_result():
    rows = _rows([5.0])
    lola._apply_fdr(rows)
    assert abs(rows[0]["qValue"] - 1e-5) < 1e-10


# ---- output.rs:335-339 test_fdr_empty
def test_fdr_empty():
    lola._apply_fdr([])


# ---- output.rs:341-366 test_fdr_identical_pvalues
def test_fdr_identical_pvalues():
    rows = _rows([3.0] * 4)
    lola._apply_fdr(rows)
    for r in rows:
        assert r["qValue"] is not None and 1e-3 - 1e-10 <= r["qValue"] <= 1.0
    assert [r["qValue"] for r in rows] == pytest.approx([1e-3] * 4, rel=1e-12)  # p * n / rank capped by the last: p


# ---- output.rs:368-407 test_fdr_preserves_order
def test_fdr_preserves_order():
    rows = _rows([10.0, 7.0, 5.0, 3.0, 2.0, 1.0, 0.5, 0.1])
    lola._apply_fdr(rows)
    srt = sorted(rows, key=lambda r: -r["pValueLog"])
    for a, b in zip(srt, srt[1:]):
        assert b["qValue"] >= a["qValue"] - 1e-10
    assert [r["qValue"] for r in rows] == oracle.bh_qvalues([r["pValueLog"] for r in rows])


# ---- output.rs:409-443 test_fdr_single_very_significant
def test_fdr_single_very_significant():
    rows = _rows([20.0, 0.1, 0.05, 0.01, 0.0])
    lola._apply_fdr(rows)
    assert rows[0]["qValue"] < 0.05
    for r in rows[1:]:
        assert r["qValue"] >= 10.0 ** (-r["pValueLog"]) - 1e-10
    assert rows[4]["qValue"] == 1.0  # p = 1


def test_fdr_infinite_pvalue_log_is_p_zero():
    rows = _rows([float("inf"), 1.0])  # output.rs:74-78: underflowed p-value
    lola._apply_fdr(rows)
    assert rows[0]["qValue"] == 0.0 and rows[1]["qValue"] == pytest.approx(0.1)


def _columns(rows):
    cols = ["userSet", "dbSet", "collection", "pValueLog", "oddsRatio", "support", "rnkPV", "rnkOR", "rnkSup", "maxRnk", "meanRnk",
            "b", "c", "d", "description", "cellType", "tissue", "antibody", "treatment", "dataSource", "filename", "qValue", "size"]
    return {c: [r.get(c) for r in rows] for c in cols}


def _full_rows(pvls):
    rows = _rows(pvls)
    for r in rows:
        r.update({"rnkPV": 1, "rnkOR": 1, "rnkSup": 1, "maxRnk": 1, "meanRnk": 1.0, "b": 5, "c": 5, "d": 100,
                  "filename": f"file{r['dbSet']}.bed", "size": 0})
    return rows


# ---- output.rs:445-465 test_write_tsv, :467-475 test_write_tsv_no_qvalue
def test_write_tsv():
    rows = _full_rows([5.0, 2.0])
    lola._apply_fdr(rows)
    buf = io.StringIO()
    lola.write_results_tsv(buf, _columns(rows))
    out = buf.getvalue()
    assert out.startswith("userSet\tdbSet\tcollection\tpValueLog\t")
    lines = out.splitlines()
    assert len(lines) == 3
    assert lines[1].startswith("1\t1\t") and lines[2].startswith("1\t2\t")  # 1-based indices
    f = lines[1].split("\t")
    assert f[3] == "5.0000" and f[4] == "1.0000" and f[10] == "1.00" and f[20] == "file0.bed"
    assert f[21] == "2.000000e-5"  # {:.6e}: Rust writes the exponent without padding (q = 1e-5 * 2 / 1)
    assert lines[2].split("\t")[21] == "1.000000e-2"
    rows = _full_rows([5.0])
    buf = io.StringIO()
    lola.write_results_tsv(buf, _columns(rows))
    assert "NA" in buf.getvalue()


# ---- enrichment.rs:498-537: odds ratio (CMLE as R's fisher.test) and its boundaries
def test_odds_ratio_kats():
    assert abs(lola.odds_ratio(10, 20, 30, 40) - 0.6693434) < 1e-3  # R: fisher.test(matrix(c(10,30,20,40), nrow=2))$estimate
    assert lola.odds_ratio(10, 0, 5, 100) == float("inf")
    assert lola.odds_ratio(0, 5, 10, 100) == 0.0


# ---- enrichment.rs:538-598: Fisher's exact test, one-sided
def test_fisher_enrichment_significant():
    assert lola.fisher_pvalue(50, 10, 5, 1000, True) < 0.001


def test_fisher_enrichment_not_significant():
    assert lola.fisher_pvalue(1, 100, 100, 1000, True) > 0.05


def test_fisher_depletion():
    assert lola.fisher_pvalue(1, 100, 100, 10, False) < 0.05


def test_fisher_edge_cases():
    assert lola.fisher_pvalue(0, 0, 0, 0, True) == 1.0
    assert lola.fisher_pvalue(0, 50, 50, 100, True) == 1.0


# ---- enrichment.rs:600-636: p_value_log
def test_p_value_log():
    pvl = lola.p_value_log(5, 15, 10, 100, True)
    assert pvl > 0.0
    assert abs(pvl - -math.log10(lola.fisher_pvalue(5, 15, 10, 100, True))) < 1e-10


def test_p_value_log_extreme():
    pvl = lola.p_value_log(50, 10, 5, 1000, True)
    assert pvl > 30.0 and math.isfinite(pvl)


def test_fisher_pvalues_against_an_independent_evaluation():
    """product (scipy.stats.hypergeom) vs oracle (pmf tail sums from lgamma): relative 1e-9 over a grid of tables,
    both directions, down to p ~ 1e-60."""
    rng = np.random.default_rng(5)
    tables = [(5, 15, 10, 100), (50, 10, 5, 1000), (1, 100, 100, 1000), (1, 100, 100, 10), (0, 5, 5, 100), (3, 0, 0, 7),
              (200, 50, 30, 5000), (1, 1, 2, 6)]
    tables += [tuple(int(v) for v in rng.integers(0, 400, 4)) for _ in range(200)]
    for a, b, c, d in tables:
        for enr in (True, False):
            got, exp = lola.fisher_pvalue(a, b, c, d, enr), oracle.fisher_pvalue(a, b, c, d, enr)
            assert got == pytest.approx(exp, rel=1e-9, abs=1e-300), (a, b, c, d, enr)


# ---- enrichment.rs:642-708 test_ranking
def test_ranking():
    rows = [{"userSet": 0, "dbSet": 0, "pValueLog": 5.0, "oddsRatio": 2.0, "support": 100},
            {"userSet": 0, "dbSet": 1, "pValueLog": 10.0, "oddsRatio": 1.0, "support": 200},
            {"userSet": 0, "dbSet": 2, "pValueLog": 3.0, "oddsRatio": 5.0, "support": 50}]
    lola._rank_results(rows)
    assert [r["rnkPV"] for r in rows] == [2, 1, 3]
    assert [r["rnkOR"] for r in rows] == [2, 3, 1]
    assert [r["rnkSup"] for r in rows] == [2, 1, 3]
    assert [r["maxRnk"] for r in rows] == [2, 3, 3]
    assert rows[0]["meanRnk"] == pytest.approx(2.0) and rows[1]["meanRnk"] == pytest.approx(5.0 / 3.0)
    assert rows[2]["meanRnk"] == pytest.approx(7.0 / 3.0)


def test_ranking_ties_and_nan():
    # ties.method = "min" (enrichment.rs:310-351), NaN odds ratios rank last and tie with each other (f64_tied)
    rows = [{"userSet": 0, "dbSet": i, "pValueLog": p, "oddsRatio": o, "support": s} for i, (p, o, s) in
            enumerate([(5.0, 2.0, 10), (5.0, float("nan"), 10), (1.0, 3.0, 20), (0.5, float("nan"), 10)])]
    lola._rank_results(rows)
    assert [r["rnkPV"] for r in rows] == [1, 1, 3, 4]
    assert [r["rnkOR"] for r in rows] == [2, 3, 1, 3]
    assert [r["rnkSup"] for r in rows] == [2, 2, 1, 2]


# ---- gtars-lola/src/universe.rs:303-347: build_restricted_universe (host-only set algebra)
def test_build_restricted_universe_kats():
    user0 = [("chr1", 100, 200), ("chr1", 300, 400)]
    user1 = [("chr1", 150, 250), ("chr2", 100, 200)]
    got = lola.build_restricted_universe([user0, user1])
    assert got == [("chr1", 100, 150), ("chr1", 150, 200), ("chr1", 200, 250), ("chr1", 300, 400), ("chr2", 100, 200)]
    assert lola.build_restricted_universe([]) == []
    assert len(lola.build_restricted_universe([[("chr1", 100, 200), ("chr1", 300, 400), ("chr1", 500, 600)]])) == 3
    # gaps are never filled, touching intervals are cut at the shared boundary (GenomicRanges disjoin)
    assert lola.build_restricted_universe([[("c", 0, 10), ("c", 10, 20), ("c", 5, 12)]]) == [
        ("c", 0, 5), ("c", 5, 10), ("c", 10, 12), ("c", 12, 20)]


def test_odds_ratio_grid_against_the_reference_method():
    """lola.odds_ratio solves the CMLE equation with its own method (Newton in log-odds); the reference uses Brent's method to an
    absolute 1e-8 in omega -- or in 1 / omega above 1, i.e. to a RELATIVE ~1e-8 * omega there (enrichment.rs:137-159).  On a
    grid of tables the two agree to that: |ours - ref| <= 2e-8 + 1e-6 * ref below 1 and <= ref * (1e-6 + 4e-8 * ref) above
    (the deliberate divergence in the last digits is stated in INTEGRATION.md); edge values are identical."""
    import math

    import oracle
    from gtars_amd.lola import odds_ratio

    rng = np.random.default_rng(12)
    tables = [(1, 1, 2, 6), (10, 5, 3, 20), (3, 0, 2, 9), (0, 4, 5, 1), (999, 1, 1, 999), (1, 999, 999, 1), (50, 50, 50, 50),
              (2, 30, 400, 5000), (120, 3, 7, 9000), (7, 7, 7, 7), (1, 0, 0, 1), (0, 0, 3, 4), (5, 5, 0, 0)]
    for _ in range(250):
        scale = int(rng.choice([5, 40, 400, 5000]))
        tables.append(tuple(int(v) for v in rng.integers(0, scale, 4)))
    worst = 0.0
    for a, b, c, d in tables:
        ours, ref = odds_ratio(a, b, c, d), oracle.odds_ratio_reference(a, b, c, d)
        if math.isnan(ref) or math.isinf(ref) or ref == 0.0:
            assert (math.isnan(ours) and math.isnan(ref)) or ours == ref, (a, b, c, d, ours, ref)
            continue
        tol = 2e-8 + 1e-6 * ref if ref <= 1.0 else ref * (1e-6 + 4e-8 * ref)
        assert abs(ours - ref) <= tol, (a, b, c, d, ours, ref)
        worst = max(worst, abs(ours - ref) / ref)
    assert worst < 1e-3


# ---------------------------------------------------------------- the compiled tail (csrc/lola_stats.cpp), all tables in one call


def _config4_like_tables(n_tables, seed, enrich):
    """Cells shaped like BASELINE config 4: universe 1e6 regions, user set 1e5, database sets hit by up to 6e4 universe regions."""
    rng = np.random.default_rng(seed)
    n_uni, n_user = 1_000_000, 100_000
    uni_hits = rng.integers(0, 60_000, n_tables)
    frac = n_user / n_uni * (rng.uniform(0.3, 4.0, n_tables) if enrich else 1.0)
    a = rng.binomial(uni_hits, np.minimum(1.0, frac))
    b = uni_hits - a
    c = n_user - a
    d = n_uni - a - b - c
    return tuple(x.astype(np.int64) for x in (a, b, c, d))


def test_compiled_tail_against_round_5_python_on_2000_tables():
    """gtars_lola_stats against the implementation it replaced (tests/lola_py_reference.py: scipy.stats.hypergeom + a numpy
    Newton solve with lgamma weights), on 2000 tables of config 4's shape, half of them enriched / depleted up to 4x.
    pValueLog: 2e-9 relative (absolute 1e-12 near p = 1; below p = 1e-300, where a double has lost its digits and scipy is off
    by tens of per cent -- 3.38e-320 for an exact 2.66e-320 --, 0.15 absolute: the exact-arithmetic test below covers that
    range).  oddsRatio: 5e-9 relative -- the OLD implementation is the
    inexact side there: its weights come from double-precision gammaln at arguments of ~1e6, 2e-9 of absolute noise each
    (the next test measures both against 60-digit arithmetic)."""
    import lola_py_reference as ref

    for enrich, direction, seed in ((False, True, 11), (True, True, 12), (True, False, 13), (False, False, 14)):
        a, b, c, d = _config4_like_tables(500, seed, enrich)
        st = lola.lola_stats(a, b, c, d, direction)
        for i in range(len(a)):
            t = (int(a[i]), int(b[i]), int(c[i]), int(d[i]))
            pvl, pvl_ref = st["pValueLog"][0, i], ref.p_value_log(*t, direction)
            tol = 0.15 if pvl_ref > 300.0 else 2e-9 * abs(pvl_ref) + 1e-12
            assert abs(pvl - pvl_ref) <= tol, (t, pvl, pvl_ref)
            orr, or_ref = st["oddsRatio"][0, i], ref.odds_ratio(*t)
            if math.isnan(or_ref) or math.isinf(or_ref) or or_ref == 0.0:
                assert (math.isnan(orr) and math.isnan(or_ref)) or orr == or_ref, (t, orr, or_ref)
            else:
                assert abs(orr - or_ref) <= 5e-9 * or_ref, (t, orr, or_ref)


def test_compiled_p_values_against_exact_arithmetic():
    """Fisher tail probabilities (enrichment.rs:19-53) against big-integer binomials / 60-digit decimals, from p ~ 0.5 down to
    the denormal range: 1e-13 relative while p is a normal double, the denormal spacing (4.94e-324) below."""
    from decimal import Decimal, getcontext
    from math import comb

    getcontext().prec = 60
    tables = [(5, 15, 10, 100), (50, 10, 5, 1000), (300, 700, 1700, 17300), (900, 100, 1100, 17900), (1200, 300, 800, 17700),
              (1990, 10, 10, 17990), (40, 960, 1960, 17040), (2, 998, 1998, 17002), (1000, 0, 1000, 18000)]
    for a, b, c, d in tables:
        n_pop, k_s, n_d = a + b + c + d, a + b, a + c
        lo, hi = max(0, k_s + n_d - n_pop), min(k_s, n_d)
        den = Decimal(comb(n_pop, n_d))
        pmf = [Decimal(comb(k_s, y) * comb(n_pop - k_s, n_d - y)) / den for y in range(lo, hi + 1)]
        for enrichment in (True, False):
            exact = sum(pmf[a - lo:]) if enrichment else sum(pmf[: a - lo + 1])
            got = lola.fisher_pvalue(a, b, c, d, enrichment)
            if a == 0 and enrichment:
                assert got == 1.0
            elif exact > Decimal("1e-300"):
                assert abs(Decimal(got) - exact) <= Decimal("1e-13") * exact, (a, b, c, d, enrichment, got, exact)
            else:
                assert abs(Decimal(got) - exact) <= Decimal("5e-324"), (a, b, c, d, enrichment, got, exact)


def test_compiled_odds_ratio_solves_the_equation_in_60_digit_arithmetic():
    """E[X | omega] = a (enrichment.rs:62-71) checked directly: the mean of the noncentral hypergeometric distribution at the
    returned omega, summed in 60-digit decimals over +-4000 terms around a, must equal a to 1e-11."""
    from decimal import Decimal, getcontext

    getcontext().prec = 60
    a_, b_, c_, d_ = _config4_like_tables(6, 21, True)
    for i in range(6):
        a, b, c, d = int(a_[i]), int(b_[i]), int(c_[i]), int(d_[i])
        m, n, k, x = a + c, b + d, a + b, a
        lo, hi = max(0, k - n), min(k, m)
        if not lo < x < hi:
            continue
        om = Decimal(lola.odds_ratio(a, b, c, d))
        w, s0, s1 = Decimal(1), Decimal(1), Decimal(0)
        for y in range(x, min(hi, x + 4000)):
            w = w * Decimal((m - y) * (k - y)) / Decimal((y + 1) * (n - k + y + 1)) * om
            s0 += w
            s1 += w * (y + 1 - x)
        w = Decimal(1)
        for y in range(x, max(lo, x - 4000), -1):
            w = w * Decimal(y * (n - k + y)) / Decimal((m - y + 1) * (k - y + 1)) / om
            s0 += w
            s1 -= w * (x - y + 1)
        assert abs(s1 / s0) < Decimal("1e-11"), (a, b, c, d, float(s1 / s0))


def test_compiled_tail_ranks_order_and_q_values_follow_the_reference_rules():
    """Ranks, the global row order and the q-values of gtars_lola_stats against the oracle's statement-level restatement of
    rank_results / the final sort / apply_fdr_correction (enrichment.rs:285-394, output.rs:35-113), fed with the library's own
    values: exact.  Three user sets; tables with a negative cell (pValueLog 0.0, oddsRatio NaN, enrichment.rs:226-247) next to
    tables with p = 1 (pValueLog -0.0: equal in the sort, NOT tied in the ranks -- f64_tied compares bits), duplicated tables
    (real ties), boundary tables (oddsRatio 0 / inf / NaN)."""
    rng = np.random.default_rng(5)
    n_db = 300
    cells = []
    for s in range(3):
        a, b, c, d = _config4_like_tables(n_db, 30 + s, True)
        a[:40], b[:40], c[:40], d[:40] = a[40:80], b[40:80], c[40:80], d[40:80]  # ties
        b[100:110] = -rng.integers(1, 5, 10)  # user regions outside the universe
        a[120:130], b[120:130] = 0, 0  # no universe region hits the set: p = 1, one-point support
        a[130:135], b[130:135], c[130:135], d[130:135] = 0, 7, 9, 50
        a[135:140], b[135:140], c[135:140], d[135:140] = 9, 0, 3, 50
        cells.append((a, b, c, d))
    a, b, c, d = (np.stack([x[j] for x in cells]) for j in range(4))
    for enrichment in (True, False):
        st = lola.lola_stats(a, b, c, d, enrichment)
        pv, orr = st["pValueLog"], st["oddsRatio"]
        assert (pv[:, 100:110] == 0.0).all() and not np.signbit(pv[:, 100:110]).any() and np.isnan(orr[:, 100:110]).all()
        if enrichment:
            assert (pv[:, 120:130] == 0.0).all() and np.signbit(pv[:, 120:130]).all()  # -log10(1 + 1e-322) = -0.0
        assert np.isnan(orr[:, 120:130]).all() and (orr[:, 130:135] == 0.0).all() and np.isinf(orr[:, 135:140]).all()
        for s in range(3):
            exp = oracle.rank_results(pv[s].tolist(), orr[s].tolist(), a[s].tolist())
            for got, e in zip((st["rnkPV"][s], st["rnkOR"][s], st["rnkSup"][s], st["maxRnk"][s]), exp[:4]):
                assert got.tolist() == e
            assert st["meanRnk"][s].tolist() == exp[4]
            if enrichment:  # the zero of a negative-cell row and the negative zero of a p = 1 row are different rank groups
                assert len({int(st["rnkPV"][s][i]) for i in (100, 120)}) == 2
        order = oracle.lola_row_order(pv.reshape(-1).tolist(), st["meanRnk"].reshape(-1).tolist())
        assert st["order"].tolist() == order
        us = [r // n_db for r in order]
        for s in range(3):
            rows = [r for r, u in zip(order, us) if u == s]
            q = oracle.bh_qvalues([pv.reshape(-1)[r] for r in rows])
            assert [st["qValue"].reshape(-1)[r] for r in rows] == q


def test_compiled_tail_argument_errors_and_values_only_form():
    from gtars_amd._lib import lib

    a, b, c, d = _config4_like_tables(16, 3, False)
    pv, orr = np.empty(16), np.empty(16)
    args = [x.ctypes.data for x in (a, b, c, d)]
    assert lib.gtars_lola_stats(*args, 16, 1, 0, pv.ctypes.data, orr.ctypes.data, *([None] * 7)) == 0
    assert pv.tolist() == [lola.p_value_log(*(int(x[i]) for x in (a, b, c, d))) for i in range(16)]
    assert lib.gtars_lola_stats(*args, 16, 1, 2, pv.ctypes.data, orr.ctypes.data, *([None] * 7)) != 0  # direction
    q = np.empty(16)
    assert lib.gtars_lola_stats(*args, 16, 1, 0, pv.ctypes.data, orr.ctypes.data, *([None] * 6), q.ctypes.data) != 0
    assert lib.gtars_lola_stats(None, *args[1:], 16, 1, 0, pv.ctypes.data, orr.ctypes.data, *([None] * 7)) != 0
