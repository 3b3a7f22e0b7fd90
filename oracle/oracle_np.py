"""Independent numpy / pure-Python literal restatement (TEST INFRASTRUCTURE ONLY).

Follows the reference's control flow one-to-one on small inputs so that the C
oracle's closed forms are checked against a second, differently-written
restatement as well as against the golden vectors.  Pure-Python loops: small cases.
"""
from itertools import combinations

import numpy as np


def r2_containment(band, curve, relax):
    """_r2_containment (_containment.py:45-80).  band: (T, j), curve: (T,)."""
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mins = np.nanmin(band, axis=1)   # pandas .min(axis=1) skips NaN (:68)
            maxs = np.nanmax(band, axis=1)   # (:69); all-NaN row -> NaN
    c = 0
    for t, val in enumerate(curve):          # (:75-77)
        if mins[t] <= val <= maxs[t]:
            c += 1
    T = len(curve)
    return c / T if relax else c // T        # (:80)


def univariate_band_depth(X, target, J=2, relax=False):
    """_univariate_band_depth (_functional.py:198-255).  X: (T, n)."""
    from scipy.special import binom
    T, n = X.shape                           # n includes the target (:229)
    others = [i for i in range(n) if i != target]   # (:235)
    depth = 0.0
    for j in range(2, J + 1):                # (:238)
        S = 0
        for seq in combinations(others, j):  # (:243-246)
            S += r2_containment(X[:, list(seq)], X[:, target], relax)   # (:251)
        depth += S / binom(n, j)             # (:253)
    return depth


def is_in_simplex_lp(simplex_points, point):
    """_is_in_simplex (_containment.py:161-176), same third-party call (scipy linprog)."""
    from scipy.optimize import linprog
    simplex_points = np.asarray(simplex_points, dtype=float)
    point = np.asarray(point, dtype=float)
    n_points = len(simplex_points)
    c = np.zeros(n_points)
    A = np.r_[simplex_points.T, np.ones((1, n_points))]
    b = np.r_[point, np.ones(1)]
    try:
        lp = linprog(c, A_eq=A, b_eq=b)
    except Exception:
        return False
    return bool(lp.success)


def pointcloud_simplex_count_lp(P, target):
    """_pointwisedepth simplex branch (_pointcloud.py:44-56) numerator, by LP."""
    n, d = P.shape
    others = [i for i in range(n) if i != target]
    return sum(is_in_simplex_lp(P[list(seq)], P[target]) for seq in combinations(others, d + 1))


def l1_depth(P, target):
    """_L1_depth (_pointcloud.py:136-150)."""
    n, d = P.shape
    e = np.zeros(d)
    for o in range(n):
        if o == target:
            continue
        with np.errstate(all="ignore"):
            e = e + (P[o] - P[target]) / np.linalg.norm(P[target] - P[o])
    return 1 - np.linalg.norm(e) / n
