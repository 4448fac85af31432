"""Multiple-testing correction for the q-value column of the result table.

Two families, as the reference offers them through getQValues (gat/Engine.pyx:2025-2040):

* the p.adjust family (gat/Stats.py:192-258, itself a re-expression of R's p.adjust): bonferroni, holm,
  hochberg, BH (= fdr, the table default, gat/__init__.py:415), BY, none; hommel raises
  NotImplementedError for more than two values exactly as the reference does;
* Storey's q-values (gat/Stats.py:26-160) with the smoother (scipy spline) and bootstrap estimates of pi0.

Everything is evaluated in the reference's operation order so that the printed q-values agree to the last
digit (tests/golden/qvalues.json).
"""
import numpy as np


class FDRResult(object):
    """what computeQValues returns (gat/Stats.py:20): qvalues, pvalues, pi0, vlambda, fdr_level, passed."""


# step-wise procedures: multiplier of the k-th p-value (k = 0 .. lp-1 in the walking order), walking order
# (ascending for the step-down procedure of Holm, descending for the step-up ones) and how the running
# extreme is taken.  `rank` below is the 1-based rank of the p-value in ASCENDING order.
def _harmonic(n):
    return np.sum(1.0 / np.arange(1, n + 1))


_STEPWISE = {
    #            descending, multiplier(n, rank)
    "holm":     (False, lambda n, rank: n - (rank - 1)),
    "hochberg": (True, lambda n, rank: n - (rank - 1)),
    "BH":       (True, lambda n, rank: float(n) / rank),
    "BY":       (True, lambda n, rank: _harmonic(n) * float(n) / rank),
}


def adjustPValues(pvalues, method="fdr", n=None):
    """adjusted p-values (gat/Stats.py:192-258).  n: number of comparisons (>= len(pvalues) in R; the
    reference asserts n <= len(pvalues), kept)."""
    if n is None:
        n = len(pvalues)
    if method == "fdr":
        method = "BH"
    p = np.array(pvalues, dtype=float)
    lp = len(p)
    assert n <= lp
    if n <= 1:
        return p
    if method == "hommel":
        if n != 2:
            raise NotImplementedError("hommel method not fully implemented")
        method = "hochberg"
    if method == "none":
        adjusted = p
    elif method == "bonferroni":
        adjusted = n * p
    elif method in _STEPWISE:
        descending, multiplier = _STEPWISE[method]
        walk = np.argsort(1 - p) if descending else np.argsort(p)
        rank = np.arange(lp, 0, -1) if descending else np.arange(1, lp + 1)
        scaled = multiplier(n, rank) * p[walk]
        running = np.minimum.accumulate(scaled) if descending else np.maximum.accumulate(scaled)
        adjusted = np.empty(lp, dtype=float)
        adjusted[walk] = running
    else:
        raise NotImplementedError("p-value adjustment method %r" % (method,))
    return np.minimum(adjusted, np.ones(lp))


def _estimate_pi0(p, vlambda, pi0_method, smooth_df, smooth_log_pi0):
    """proportion of true null hypotheses (gat/Stats.py:49-108)."""
    m = len(p)
    if isinstance(vlambda, float):
        vlambda = (vlambda,)
    nl = len(vlambda)
    if 1 < nl < 4:
        raise ValueError(" if length of vlambda greater than 1, you need at least 4 values.")
    if nl > 1 and (min(vlambda) < 0 or max(vlambda) >= 1):
        raise ValueError("vlambda must be within [0, 1).")
    if nl == 1:
        lam = vlambda[0]
        if lam < 0 or lam >= 1:
            raise ValueError("vlambda must be within [0, 1).")
        return min(np.mean(p >= lam) / (1.0 - lam), 1.0), lam
    lams = np.asarray(vlambda, dtype=float)
    at_least = np.array([np.mean(p >= lam) for lam in lams]) / (1.0 - lams)
    if pi0_method == "smoother":
        import scipy.interpolate
        y = np.log(at_least) if smooth_log_pi0 else at_least
        tck = scipy.interpolate.splrep(lams, y, k=smooth_df, s=10000)
        pi0 = scipy.interpolate.splev(max(lams), tck)
        if smooth_log_pi0:
            pi0 = np.exp(pi0)
    elif pi0_method == "bootstrap":
        floor = min(at_least)
        mse = np.zeros(nl, dtype=float)
        for _ in range(100):
            # numpy.random.random_integers(0, m - 1, m) of the reference: same draws from the global stream
            boot = p[np.random.randint(0, m, m)]
            above = np.array([np.mean(boot > lam) for lam in lams]) / (1.0 - lams)
            mse += (above - floor) ** 2
        pi0 = min(at_least[mse == min(mse)])
    else:
        raise ValueError("'pi0_method' must be one of 'smoother' or 'bootstrap'.")
    return min(pi0, 1.0), vlambda


def computeQValues(pvalues, vlambda=None, pi0_method="smoother", fdr_level=None, robust=False, smooth_df=3,
                   smooth_log_pi0=False, pi0=None):
    """q-values after Storey (2002) (gat/Stats.py:26-160)."""
    if min(pvalues) < 0 or max(pvalues) > 1:
        raise ValueError("p-values out of range")
    p = np.array(pvalues, dtype=float)
    m = len(p)
    if vlambda is None:
        vlambda = np.arange(0, 0.95, 0.05)
    if pi0 is None:
        pi0, vlambda = _estimate_pi0(p, vlambda, pi0_method, smooth_df, smooth_log_pi0)
    if pi0 <= 0:
        raise ValueError("The estimated pi0 <= 0 (%f). Check that you have valid p-values or use another vlambda method." % pi0)
    if fdr_level is not None and (fdr_level <= 0 or fdr_level > 1):
        raise ValueError("'fdr_level' must be within (0, 1].")
    ascending = np.argsort(p)
    ordered = p[ascending]
    at_most = np.searchsorted(ordered, p, side="right")      # v[i] = #{j: p_j <= p_i} (gat/Stats.py:123-133)
    q = p * pi0 * m / at_most
    if robust:
        q /= (1.0 - (1.0 - p) ** m)
    # bounded by 1 and monotone in p (gat/Stats.py:139-142)
    capped = np.minimum(np.minimum.accumulate(q[ascending][::-1])[::-1], 1.0)
    q = np.empty(m, dtype=float)
    q[ascending] = capped
    result = FDRResult()
    result.qvalues = q
    result.passed = [x <= fdr_level for x in q] if fdr_level is not None else [False for _ in q]
    result.pvalues = p
    result.pi0 = pi0
    result.vlambda = vlambda
    result.fdr_level = fdr_level
    return result


def getQValues(pvalues, method="storey", **kwargs):
    """gat/Engine.pyx:2025-2040.  A failing Storey estimate (ValueError) falls back to q = 1 for every row, as
    there.  Without a `vlambda` keyword the reference hands computeQValues the default grid as an array and its
    `vlambda == None` test raises under numpy >= 1.13 (-> all 1.0); the grid is used here, which is what the
    command line (vlambda=None, gat/IO.py:474-477) gets in the reference too."""
    if method == "storey":
        try:
            fdr = computeQValues(pvalues, vlambda=kwargs.get("vlambda", None),
                                 pi0_method=kwargs.get("pi0_method", "smoother"))
        except ValueError as msg:
            import logging
            logging.getLogger("gat").warning("qvalue computation failed: %s" % msg)
            return [1.0] * len(pvalues)
        return fdr.qvalues
    return adjustPValues(pvalues, method=method)
