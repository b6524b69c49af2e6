// lola_stats.cpp -- the statistics tail of run_lola in compiled host code behind the C ABI
// (include/gtars_amd_host.h: gtars_lola_stats): Fisher's exact p-value, the conditional-MLE odds ratio, min-ranks, the
// global row order and Benjamini-Hochberg q-values for every (user set, database set) table of a run, threaded over the
// tables.  Rules: gtars-lola/src/enrichment.rs:19-169 (values), 243-294 (negative cells, global order), 296-394 (ranks),
// gtars-lola/src/output.rs:35-113 (FDR).  Round 5 walked the tables one by one in Python (scipy + a numpy Newton solve per
// table): 15 s for config 4's 2000 tables behind a 0.2-ms count step.
//
// Numerics.  The reference sums hypergeometric probabilities through statrs 0.18 (third party, not in the checkout) over the
// whole tail, and finds the odds ratio by Brent's method over the whole support of the table -- O(support) per evaluation,
// i.e. 1e5 terms per table for a 1e5-region user set.  Both distributions are unimodal and their terms fall off faster than
// geometrically away from the mode, so everything here is summed over the WINDOW of terms that can still change a double
// (relative 1e-22), by the exact term ratio  pmf(y+1) / pmf(y) = (K-y)(n-y) / ((y+1)(N-K-n+y+1))  -- no lgamma noise
// between neighbouring terms; the one anchor a p-value needs (the tail's largest term relative to the distribution's mode)
// is a difference of lgammal values in 80-bit arithmetic (at N = 1e6 a double lgamma carries 2e-9 of absolute error, which
// IS the relative error of the probability).  p = (tail sum) / (sum over the support), both relative to the mode.
// The integer cells, ranks and q-values-given-p-values follow the reference's rules exactly; pValueLog / oddsRatio agree
// with the reference to floating-point tolerance only (statrs unpinned; the reference's own Brent solve stops
// at an absolute 1e-8 in omega).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gtars_amd_host.h"

namespace gtars {
gtars_status fail(gtars_status st, const std::string &msg);
const char *cfg_get(const char *name);
}
extern "C" uint32_t gtars_host_threads(uint32_t cap);

namespace {

typedef long double ld;

// ln C(n, k) in 80-bit arithmetic
inline ld ln_choose(uint64_t n, uint64_t k) { return lgammal((ld)n + 1) - lgammal((ld)k + 1) - lgammal((ld)(n - k) + 1); }

// Central hypergeometric distribution: population N, K success states, n draws; support [lo, hi].
struct Hyper {
    uint64_t N, K, n, lo, hi;
    // pmf(y + 1) / pmf(y)
    inline double up(uint64_t y) const { return ((double)(K - y) * (double)(n - y)) / ((double)(y + 1) * (double)(N - K - n + y + 1)); }
    uint64_t mode() const {
        // floor((n + 1)(K + 1) / (N + 2)), clamped to the support
        const unsigned __int128 num = (unsigned __int128)(n + 1) * (K + 1);
        uint64_t m = (uint64_t)(num / (N + 2));
        return std::min(std::max(m, lo), hi);
    }
    // sum of pmf(y) / pmf(anchor) over y in [from, to] (anchor inside or at an end of that range is the caller's business);
    // walks outward from `start` in both directions until a term no longer matters
    double window_sum(uint64_t start, uint64_t from, uint64_t to) const {
        // the four factors of the term ratio carried as doubles and stepped by one (exact below 2^53)
        double s = 1.0, w = 1.0;
        double fa = (double)(K - start), fb = (double)(n - start), fc = (double)(start + 1), fd = (double)(N - K - n + start + 1);
        for (uint64_t y = start; y < to; ++y) {  // upwards
            const double r = (fa * fb) / (fc * fd);
            w *= r;
            s += w;
            fa -= 1.0, fb -= 1.0, fc += 1.0, fd += 1.0;
            if (w < 1e-19 * s && r < 1.0) break;  // the ratio descends with y: from here on a geometric tail
        }
        w = 1.0;
        fa = (double)(K - start), fb = (double)(n - start), fc = (double)(start + 1), fd = (double)(N - K - n + start + 1);
        for (uint64_t y = start; y > from; --y) {  // downwards
            fa += 1.0, fb += 1.0, fc -= 1.0, fd -= 1.0;
            const double r = (fc * fd) / (fa * fb);
            w *= r;
            s += w;
            if (w < 1e-19 * s && r < 1.0) break;
        }
        return s;
    }
    // ln(pmf(y) / pmf(m))
    ld ln_ratio(uint64_t y, uint64_t m) const {
        if (y == m) return 0;
        const uint64_t d = y > m ? y - m : m - y;
        if (d <= 64) {  // a short product of exact ratios beats the lgamma difference
            ld r = 0;
            if (y > m)
                for (uint64_t t = m; t < y; ++t) r += logl(((ld)(K - t) * (ld)(n - t)) / ((ld)(t + 1) * (ld)(N - K - n + t + 1)));
            else
                for (uint64_t t = y; t < m; ++t) r -= logl(((ld)(K - t) * (ld)(n - t)) / ((ld)(t + 1) * (ld)(N - K - n + t + 1)));
            return r;
        }
        return (ln_choose(K, y) + ln_choose(N - K, n - y)) - (ln_choose(K, m) + ln_choose(N - K, n - m));
    }
    // ln P(from <= X <= to)
    ld ln_prob(uint64_t from, uint64_t to) const {
        const uint64_t m = mode();
        const ld total = logl((ld)window_sum(m, lo, hi));
        const uint64_t a = std::min(std::max(m, from), to);  // the range's largest term
        return ln_ratio(a, m) + logl((ld)window_sum(a, from, to)) - total;
    }
};

// ContingencyTable::fisher_pvalue (enrichment.rs:19-53)
double fisher_pvalue(uint64_t a, uint64_t b, uint64_t c, uint64_t d, bool enrichment) {
    const uint64_t n_pop = a + b + c + d, k_success = a + b, n_draws = a + c;
    if (n_pop == 0 || k_success == 0 || n_draws == 0) return 1.0;
    if (k_success > n_pop || n_draws > n_pop) return 1.0;
    Hyper h{n_pop, k_success, n_draws, 0, 0};
    h.lo = k_success + n_draws > n_pop ? k_success + n_draws - n_pop : 0;
    h.hi = std::min(k_success, n_draws);
    if (enrichment) {
        if (a == 0) return 1.0;
        // sf(a - 1) = P(X >= a): 1 below the support, 0 at or above its top (statrs: x < min -> 1, x >= max -> 0)
        if (a - 1 < h.lo) return 1.0;
        if (a - 1 >= h.hi) return 0.0;
        const ld lp = h.ln_prob(a, h.hi);
        return lp >= 0 ? 1.0 : (double)expl(lp);
    }
    // cdf(a) = P(X <= a)
    if (a < h.lo) return 0.0;
    if (a >= h.hi) return 1.0;
    const ld lp = h.ln_prob(h.lo, a);
    return lp >= 0 ? 1.0 : (double)expl(lp);
}

// ContingencyTable::odds_ratio (enrichment.rs:62-160): the omega for which Fisher's noncentral hypergeometric distribution of
// the table's margins has mean a.  Solved in theta = ln(omega), where the mean is strictly increasing with the variance as its
// derivative: Newton steps kept inside a sign-change bracket.  Mean and variance at a theta are summed over the window around
// the TILTED distribution's mode by the term ratio  up(y) * omega.
struct Tilted {
    uint64_t m, n, k, lo, hi;  // column totals m = a + c, n = b + d; row total k = a + b
    inline double up(uint64_t y) const { return ((double)(m - y) * (double)(k - y)) / ((double)(y + 1) * (double)(n - k + y + 1)); }
    // the y in [lo, hi] with the largest weight at omega: the last y with up(y - 1) * omega >= 1 (up() descends)
    uint64_t mode(double omega) const {
        uint64_t l = lo, h = hi;  // invariant: weight ascends up to l; find the first y in (lo, hi] with up(y - 1) * omega < 1
        while (l < h) {
            const uint64_t mid = l + (h - l + 1) / 2;
            if (up(mid - 1) * omega >= 1.0) l = mid; else h = mid - 1;
        }
        return l;
    }
    void moments(double theta, double &mu, double &var) const {
        const double omega = std::exp(theta);
        const uint64_t c = mode(omega);
        // sums of w, w * (y - c), w * (y - c)^2 relative to the mode's weight; the four factors of the term ratio are carried as
        // doubles and stepped by one (exact below 2^53), so a term costs one division off the dependent chain
        double s0 = 1.0, s1 = 0.0, s2 = 0.0, w = 1.0, t = 0.0;
        double fa = (double)(m - c), fb = (double)(k - c), fc = (double)(c + 1), fd = (double)(n - k + c + 1);
        for (uint64_t y = c; y < hi; ++y) {
            w *= (fa * fb * omega) / (fc * fd);
            fa -= 1.0, fb -= 1.0, fc += 1.0, fd += 1.0, t += 1.0;
            const double wt = w * t, wtt = wt * t;
            s0 += w, s1 += wt, s2 += wtt;
            if (wtt < 1e-19 * s0 && t > 8.0) break;
        }
        w = 1.0, t = 0.0;
        fa = (double)(m - c), fb = (double)(k - c), fc = (double)(c + 1), fd = (double)(n - k + c + 1);
        const double inv_omega = 1.0 / omega;
        for (uint64_t y = c; y > lo; --y) {
            fa += 1.0, fb += 1.0, fc -= 1.0, fd -= 1.0, t += 1.0;
            w *= (fc * fd * inv_omega) / (fa * fb);
            const double wt = w * t, wtt = wt * t;
            s0 += w, s1 -= wt, s2 += wtt;
            if (wtt < 1e-19 * s0 && t > 8.0) break;
        }
        const double e1 = s1 / s0;
        mu = (double)c + e1;
        var = s2 / s0 - e1 * e1;
    }
};

double odds_ratio(uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
    Tilted t{a + c, b + d, a + b, 0, 0};
    const uint64_t x = a;
    t.lo = t.k > t.n ? t.k - t.n : 0;
    t.hi = std::min(t.k, t.m);
    if (t.lo == t.hi) return std::numeric_limits<double>::quiet_NaN();
    if (x == t.lo) return 0.0;
    if (x == t.hi) return std::numeric_limits<double>::infinity();
    const double target = (double)x, inf = std::numeric_limits<double>::infinity();
    double mu, var;
    t.moments(0.0, mu, var);
    if (std::fabs(mu - target) < 1e-12) return 1.0;  // (enrichment.rs:140-142)
    // The mean runs from lo to hi as theta goes from -inf to +inf and lo < x < hi: a unique root.  Newton from the sample log odds
    // ratio (which the conditional estimate approaches for large tables) inside a sign-change bracket that starts unbounded on
    // the side not yet seen; a step that is not finite or longer than 8 (flat tail of the mean) is cut to 8, one that would
    // leave the bracket is replaced by the bracket's middle.
    double t_lo = -inf, t_hi = inf;
    if (mu < target) t_lo = 0.0; else t_hi = 0.0;
    double theta = std::log(((double)a + 0.5) * ((double)d + 0.5) / (((double)b + 0.5) * ((double)c + 0.5)));
    if (!(t_lo < theta && theta < t_hi)) theta = mu < target ? 1.0 : -1.0;
    for (int it = 0; it < 200; ++it) {
        t.moments(theta, mu, var);
        if (mu == target) break;
        if (mu < target) t_lo = theta; else t_hi = theta;
        double step = var > 0.0 ? -(mu - target) / var : std::numeric_limits<double>::quiet_NaN();
        if (!(std::fabs(step) <= 8.0)) step = mu < target ? 8.0 : -8.0;
        double nxt = theta + step;
        const double scale = std::max(1.0, std::fabs(theta));
        if (std::fabs(nxt - theta) <= 1e-13 * scale) {  // (a step below the resolution of theta included: nxt == theta)
            theta = nxt;
            break;
        }
        // a step past the far end of the bracket (that end is finite: the step points at it, away from theta's own side)
        if (!(t_lo < nxt && nxt < t_hi)) nxt = 0.5 * (t_lo + t_hi);
        theta = nxt;
        if (t_hi - t_lo <= 1e-14 * scale) break;
    }
    return std::exp(theta);
}

// f64_tied (enrichment.rs:296-303): NaN ties with NaN, everything else by its bits (0.0 and -0.0 are NOT tied)
inline bool f64_tied(double a, double b) {
    if (std::isnan(a) && std::isnan(b)) return true;
    uint64_t x, y;
    memcpy(&x, &a, 8), memcpy(&y, &b, 8);
    return x == y;
}

// partial_cmp(...).unwrap_or(Equal) as a strict-weak "a sorts in front of b" for a DESCENDING order: NaN compares Equal to
// everything, exactly as the reference's comparator does (std::stable_sort is a merge sort like Rust's sort_by; with a
// comparator that is not a strict weak order in the presence of NaN the two may place NaN rows differently -- the reference's
// own order is unspecified there; pValueLog is never NaN on this path)
inline bool desc_before(double a, double b) { return a > b; }

// f(i) for i in [0, n) on the process's host threads, `chunk` indices at a time (tables differ a lot in cost: small dynamic chunks)
template <class F>
void parallel_for(size_t n, size_t chunk, F &&f) {
    const unsigned nt = (unsigned)std::min<size_t>(gtars_host_threads(64), (n + chunk - 1) / chunk);
    if (nt <= 1) {
        for (size_t i = 0; i < n; ++i) f(i);
        return;
    }
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    auto body = [&]() {
        for (;;) {
            const size_t i0 = next.fetch_add(chunk);
            if (i0 >= n) return;
            for (size_t i = i0; i < std::min(n, i0 + chunk); ++i) f(i);
        }
    };
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(body);
    body();
    for (auto &x : th) x.join();
}

// rank_results on one user set's rows (enrichment.rs:353-394): ONE index vector re-sorted three times (stable sorts: ties keep
// the previous sort's order, as the reference's do), min-ranks, then the combined ranks
void rank_rows(const double *pv, const double *orr, const uint64_t *sup, size_t n, uint32_t *rnk_pv, uint32_t *rnk_or,
               uint32_t *rnk_sup, uint32_t *max_rnk, double *mean_rnk) {
    std::vector<uint32_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0u);
    auto min_ranks = [&](uint32_t *out, auto tied) {  // assign_min_ranks_* (enrichment.rs:305-351)
        uint32_t rank = 1;
        for (size_t i = 0; i < n; ++i) {
            if (i > 0 && !tied(idx[i - 1], idx[i])) rank = (uint32_t)i + 1;
            out[idx[i]] = rank;
        }
    };
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return desc_before(pv[x], pv[y]); });
    min_ranks(rnk_pv, [&](uint32_t x, uint32_t y) { return f64_tied(pv[x], pv[y]); });
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) {  // descending, NaN last
        const bool nx = std::isnan(orr[x]), ny = std::isnan(orr[y]);
        if (nx || ny) return !nx && ny;
        return orr[x] > orr[y];
    });
    min_ranks(rnk_or, [&](uint32_t x, uint32_t y) { return f64_tied(orr[x], orr[y]); });
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return sup[x] > sup[y]; });
    min_ranks(rnk_sup, [&](uint32_t x, uint32_t y) { return sup[x] == sup[y]; });
    for (size_t f = 0; f < n; ++f) {
        const uint32_t p = rnk_pv[f], o = rnk_or[f], r = rnk_sup[f];
        max_rnk[f] = std::max(p, std::max(o, r));
        mean_rnk[f] = (double)((uint64_t)p + o + r) / 3.0;
    }
}

// Benjamini-Hochberg over one user set's rows `idx` (positions in p_value_log / q_value, in the order the rows stand in the
// result list): output.rs:62-112
void bh_rows(const double *p_value_log, std::vector<uint64_t> &idx, double *q_value) {
    const size_t n = idx.size();
    if (!n) return;
    std::stable_sort(idx.begin(), idx.end(), [&](uint64_t x, uint64_t y) { return desc_before(p_value_log[x], p_value_log[y]); });
    std::vector<double> p(n);
    for (size_t i = 0; i < n; ++i) {
        const double pvl = p_value_log[idx[i]];
        p[i] = pvl == std::numeric_limits<double>::infinity() ? 0.0 : std::pow(10.0, -pvl);
    }
    double q = std::min(p[n - 1] * (double)n / (double)n, 1.0);
    q_value[idx[n - 1]] = q;
    for (size_t i = n - 1; i-- > 0;) {
        q = std::min(std::min(p[i] * (double)n / (double)(i + 1), q), 1.0);
        q_value[idx[i]] = q;
    }
}

template <class F>
gtars_status guarded(F &&f) {
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return gtars::fail(GTARS_ERR_INTERNAL, "out of host memory");
    } catch (const std::exception &e) {
        return gtars::fail(GTARS_ERR_INTERNAL, std::string("internal error: ") + e.what());
    }
}

}  // namespace

extern "C" gtars_status gtars_lola_rank(const double *p_value_log, const double *odds, const uint64_t *support, uint64_t n,
                                        uint32_t *rnk_pv, uint32_t *rnk_or, uint32_t *rnk_sup, uint32_t *max_rnk,
                                        double *mean_rnk) {
    if (n && (!p_value_log || !odds || !support || !rnk_pv || !rnk_or || !rnk_sup || !max_rnk || !mean_rnk))
        return gtars::fail(GTARS_ERR_INVALID_ARG, "gtars_lola_rank: null argument");
    if (n >= 0xFFFFFFFFull) return gtars::fail(GTARS_ERR_INVALID_ARG, "gtars_lola_rank: too many rows");
    return guarded([&]() -> gtars_status {
        rank_rows(p_value_log, odds, support, (size_t)n, rnk_pv, rnk_or, rnk_sup, max_rnk, mean_rnk);
        return GTARS_OK;
    });
}

extern "C" gtars_status gtars_lola_fdr(const double *p_value_log, const uint64_t *user_set, uint64_t n_rows, double *q_value) {
    if (n_rows && (!p_value_log || !user_set || !q_value)) return gtars::fail(GTARS_ERR_INVALID_ARG, "gtars_lola_fdr: null argument");
    return guarded([&]() -> gtars_status {
        // apply_fdr_correction (output.rs:35-113): the rows of every user set, in the order they stand
        std::vector<uint64_t> sets(user_set, user_set + n_rows);
        std::sort(sets.begin(), sets.end());
        sets.erase(std::unique(sets.begin(), sets.end()), sets.end());
        std::vector<std::vector<uint64_t>> by_set(sets.size());
        for (uint64_t r = 0; r < n_rows; ++r)
            by_set[(size_t)(std::lower_bound(sets.begin(), sets.end(), user_set[r]) - sets.begin())].push_back(r);
        for (auto &idx : by_set) bh_rows(p_value_log, idx, q_value);
        return GTARS_OK;
    });
}

extern "C" gtars_status gtars_lola_stats(const int64_t *a, const int64_t *b, const int64_t *c, const int64_t *d, uint64_t n_db,
                                         uint64_t n_user_sets, int direction, double *p_value_log, double *odds,
                                         uint32_t *rnk_pv, uint32_t *rnk_or, uint32_t *rnk_sup, uint32_t *max_rnk,
                                         double *mean_rnk, uint64_t *order, double *q_value) {
    const bool want_ranks = rnk_pv && rnk_or && rnk_sup && max_rnk && mean_rnk;
    if (!a || !b || !c || !d || !p_value_log || !odds) return gtars::fail(GTARS_ERR_INVALID_ARG, "gtars_lola_stats: null argument");
    if (direction != 0 && direction != 1) return gtars::fail(GTARS_ERR_INVALID_ARG, "gtars_lola_stats: direction must be 0 (enrichment) or 1 (depletion)");
    if (n_db >= 0xFFFFFFFFull) return gtars::fail(GTARS_ERR_INVALID_ARG, "gtars_lola_stats: too many database sets");
    if (!want_ranks && (rnk_pv || rnk_or || rnk_sup || max_rnk || mean_rnk || order || q_value))
        return gtars::fail(GTARS_ERR_INVALID_ARG, "gtars_lola_stats: the five rank outputs come together, and order / q_value need them");
    if (q_value && !order) return gtars::fail(GTARS_ERR_INVALID_ARG, "gtars_lola_stats: q_value needs the order output");
    return guarded([&]() -> gtars_status {
        const size_t total = (size_t)(n_db * n_user_sets);
        const bool enrichment = direction == 0;
        // values (enrichment.rs:226-247): a table with a negative cell gets pValueLog 0 and oddsRatio NaN
        parallel_for(total, 8, [&](size_t i) {
            if (a[i] < 0 || b[i] < 0 || c[i] < 0 || d[i] < 0) {
                p_value_log[i] = 0.0;
                odds[i] = std::numeric_limits<double>::quiet_NaN();
                return;
            }
            const uint64_t ua = (uint64_t)a[i], ub = (uint64_t)b[i], uc = (uint64_t)c[i], ud = (uint64_t)d[i];
            p_value_log[i] = -std::log10(fisher_pvalue(ua, ub, uc, ud, enrichment) + 1e-322);  // enrichment.rs:166-169
            odds[i] = odds_ratio(ua, ub, uc, ud);
        });
        if (!want_ranks) return GTARS_OK;
        parallel_for((size_t)n_user_sets, 1, [&](size_t us) {
            const size_t base = us * (size_t)n_db;
            // support is u64 in the reference (a count: never negative)
            rank_rows(p_value_log + base, odds + base, reinterpret_cast<const uint64_t *>(a) + base, (size_t)n_db, rnk_pv + base,
                      rnk_or + base, rnk_sup + base, max_rnk + base, mean_rnk + base);
        });
        if (!order) return GTARS_OK;
        // global order (enrichment.rs:285-294): pValueLog descending, then meanRnk ascending, stable over (user set, db set)
        std::iota(order, order + total, (uint64_t)0);
        std::stable_sort(order, order + total, [&](uint64_t x, uint64_t y) {
            if (p_value_log[x] > p_value_log[y]) return true;
            if (p_value_log[x] < p_value_log[y]) return false;
            return mean_rnk[x] < mean_rnk[y];
        });
        if (!q_value) return GTARS_OK;
        // apply_fdr_correction (output.rs:35-113) on the rows in that order: Benjamini-Hochberg per user set
        std::vector<std::vector<uint64_t>> by_set((size_t)n_user_sets);
        for (auto &v : by_set) v.reserve((size_t)n_db);
        for (size_t r = 0; r < total; ++r) by_set[(size_t)(order[r] / n_db)].push_back(order[r]);
        parallel_for((size_t)n_user_sets, 1, [&](size_t us) { bh_rows(p_value_log, by_set[us], q_value); });
        return GTARS_OK;
    });
}

// One table (the scalar functions of the reference's ContingencyTable, for bindings and tests)
extern "C" double gtars_lola_fisher_pvalue(uint64_t a, uint64_t b, uint64_t c, uint64_t d, int direction) {
    return fisher_pvalue(a, b, c, d, direction == 0);
}
extern "C" double gtars_lola_odds_ratio(uint64_t a, uint64_t b, uint64_t c, uint64_t d) { return odds_ratio(a, b, c, d); }
