"""Round 5's statistics tail of run_lola, kept as a TEST reference (it was ``gtars_amd/lola.py`` until the tail moved into the
library's compiled code, csrc/lola_stats.cpp): Fisher p-values through scipy.stats.hypergeom and the conditional-MLE odds
ratio by a numpy Newton solve over the table's whole support with lgamma weights.  Not imported by the product."""
import math

import numpy as np


def fisher_pvalue(a: int, b: int, c: int, d: int, enrichment: bool = True) -> float:
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
    return -math.log10(fisher_pvalue(a, b, c, d, enrichment) + 1e-322)


def odds_ratio(a: int, b: int, c: int, d: int) -> float:
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
        if not (t_lo < nxt < t_hi):
            nxt = 0.5 * (t_lo + t_hi)
        if abs(nxt - theta) <= 1e-13 * max(1.0, abs(theta)) or t_hi - t_lo <= 1e-14 * max(1.0, abs(theta)):
            theta = nxt
            break
        theta = nxt
    return math.exp(theta)
