"""p-value adjustment (gat/Stats.py:192-258, a re-expression of R's p.adjust) for the q-value
column of the result table (default method BH, gat/__init__.py:415)."""
import numpy as np


def adjustPValues(pvalues, method="fdr", n=None):
    if n is None:
        n = len(pvalues)
    if method == "fdr":
        method = "BH"
    p = np.array(pvalues, dtype=float)
    lp = len(p)
    assert n <= lp
    if n <= 1:
        return p
    if method == "bonferroni":
        p0 = n * p
    elif method == "holm":
        i = np.arange(lp)
        o = np.argsort(p)
        ro = np.argsort(o)
        p0 = np.maximum.accumulate((n - i) * p[o])[ro]
    elif method == "hochberg":
        i = np.arange(0, lp)[::-1]
        o = np.argsort(1 - p)
        ro = np.argsort(o)
        p0 = np.minimum.accumulate((n - i) * p[o])[ro]
    elif method == "BH":
        i = np.arange(1, lp + 1)[::-1]
        o = np.argsort(1 - p)
        ro = np.argsort(o)
        p0 = np.minimum.accumulate(float(n) / i * p[o])[ro]
    elif method == "BY":
        i = np.arange(1, lp + 1)[::-1]
        o = np.argsort(1 - p)
        ro = np.argsort(o)
        q = np.sum(1.0 / np.arange(1, n + 1))
        p0 = np.minimum.accumulate(q * float(n) / i * p[o])[ro]
    elif method == "none":
        p0 = p
    else:
        raise NotImplementedError("p-value adjustment method %r" % method)
    return np.minimum(p0, np.ones(len(p0)))


def getQValues(pvalues, method="BH", **kwargs):
    """gat/Engine.pyx:2025-2040 (storey's method is outside the accelerated path's scope)."""
    if method == "storey":
        raise NotImplementedError("qvalue method 'storey' is not implemented; use BH (the default), BY, holm, ...")
    return adjustPValues(pvalues, method=method)
