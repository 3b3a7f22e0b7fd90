"""CPU oracle for the statdepth band-depth hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (statdepth_amd/) never does.  Parity status: PINNED
against golden vectors produced by the reference itself (tests/golden/).

`oracle.c` (C, OpenMP) is the restatement; this module is its ctypes binding plus
the float normalisers of the reference (file:line cited per function).
`oracle_np` is an independent numpy / pure-Python literal restatement used to
cross-check the C one on small cases.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        c_dp = ctypes.POINTER(ctypes.c_double)
        c_lp = ctypes.POINTER(ctypes.c_long)
        c_ip = ctypes.POINTER(ctypes.c_int64)
        L.oracle_band_enum.argtypes = [c_dp, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_long,
                                       c_lp, ctypes.c_long, ctypes.c_int, ctypes.c_int, c_ip]
        L.oracle_multi_band_enum.argtypes = [c_dp, ctypes.c_long, ctypes.c_long, ctypes.c_int, c_lp, ctypes.c_long,
                                             ctypes.c_int, ctypes.c_int, c_ip]
        L.oracle_mbd_counts.argtypes = [c_dp, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_long,
                                        c_lp, ctypes.c_long, ctypes.c_int, c_ip]
        L.oracle_mbd_counts_ranksort.argtypes = [c_dp, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_long,
                                                 ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]
        L.oracle_above_below.argtypes = [c_dp, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_long,
                                         c_lp, ctypes.c_long, c_ip]
        L.oracle_bd_strict_counts.argtypes = [c_dp, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_long,
                                              c_lp, ctypes.c_long, c_ip]
        L.oracle_point_in_hull.argtypes = [c_dp, ctypes.c_int, ctypes.c_int, c_dp, ctypes.c_double]
        L.oracle_pointcloud_simplex_counts.argtypes = [c_dp, ctypes.c_long, ctypes.c_int, c_lp, ctypes.c_long,
                                                       ctypes.c_double, c_ip]
        L.oracle_multi_simplex_counts.argtypes = [c_dp, ctypes.c_long, ctypes.c_long, ctypes.c_int, c_lp,
                                                  ctypes.c_long, ctypes.c_int, ctypes.c_double, c_ip]
        L.oracle_simplex_sampled.argtypes = [c_dp, ctypes.c_long, ctypes.c_long, ctypes.c_int, c_lp, ctypes.c_long,
                                             ctypes.c_int, ctypes.c_double, ctypes.c_long, ctypes.c_uint64, c_ip]
        L.oracle_l1_depth.argtypes = [c_dp, ctypes.c_long, ctypes.c_int, c_lp, ctypes.c_long, c_dp]
        L.oracle_num_threads.restype = ctypes.c_int
        L.oracle_set_num_threads.argtypes = [ctypes.c_int]
        _LIB = L
    return _LIB


DEFAULT_TOL = 1e-7   # feasibility tolerance of the LP the reference calls (_containment.py:171; HiGHS default)


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _lp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_long))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


def _x2d(X):
    """(T, n) float64 array in either pandas layout -> (array kept alive, st, sn)."""
    X = np.asarray(X, dtype=np.float64)
    assert X.ndim == 2
    if not (X.flags.c_contiguous or X.flags.f_contiguous):
        X = np.ascontiguousarray(X)
    st, sn = X.strides[0] // 8, X.strides[1] // 8
    return X, st, sn


def _targets(targets, n):
    if targets is None:
        targets = np.arange(n)
    t = np.ascontiguousarray(np.asarray(targets, dtype=np.int64))
    assert t.ndim == 1 and (len(t) == 0 or (t.min() >= 0 and t.max() < n))
    return t


def num_threads():
    return int(lib().oracle_num_threads())


def set_num_threads(k):
    """Threads used by the OpenMP loops (bench.py pins this to the box's CPU share)."""
    lib().oracle_set_num_threads(int(k))
    return num_threads()


def band_enum(X, targets=None, J=2, relax=True):
    """Literal subset enumeration (_functional.py:238-253): int64[m, J-1]."""
    X, st, sn = _x2d(X)
    T, n = X.shape
    tg = _targets(targets, n)
    out = np.zeros((len(tg), J - 1), dtype=np.int64)
    rc = lib().oracle_band_enum(_dp(X), T, n, st, sn, _lp(tg), len(tg), J, int(bool(relax)), _ip(out))
    assert rc == 0
    return out


def multi_band_enum(P, targets=None, J=2, relax=True):
    """'r2_enum' componentwise band containment of multivariate curves P (n, T, d), literal enumeration: int64[m, J-1]."""
    P = np.ascontiguousarray(np.asarray(P, dtype=np.float64))
    n, T, d = P.shape
    tg = _targets(targets, n)
    out = np.zeros((len(tg), J - 1), dtype=np.int64)
    rc = lib().oracle_multi_band_enum(_dp(P), n, T, d, _lp(tg), len(tg), J, int(bool(relax)), _ip(out))
    assert rc == 0
    return out


def mbd_counts(X, targets=None, J=2):
    """Closed-form relax=True totals: int64[m, J-1] = sum_t #contained j-bands."""
    X, st, sn = _x2d(X)
    T, n = X.shape
    tg = _targets(targets, n)
    out = np.zeros((len(tg), J - 1), dtype=np.int64)
    rc = lib().oracle_mbd_counts(_dp(X), T, n, st, sn, _lp(tg), len(tg), J, _ip(out))
    assert rc == 0
    return out


def mbd_counts_ranksort(X, J=2):
    """The same totals for ALL curves by per-timepoint sorting (the rank formulation): int64[n, J-1]."""
    X, st, sn = _x2d(X)
    T, n = X.shape
    out = np.zeros((n, J - 1), dtype=np.int64)
    rc = lib().oracle_mbd_counts_ranksort(_dp(X), T, n, st, sn, J, _ip(out))
    assert rc == 0
    return out


def above_below(X, targets=None):
    """int64[m, T, 2]: strictly-above / strictly-below counts per (target, t)."""
    X, st, sn = _x2d(X)
    T, n = X.shape
    tg = _targets(targets, n)
    out = np.zeros((len(tg), T, 2), dtype=np.int64)
    rc = lib().oracle_above_below(_dp(X), T, n, st, sn, _lp(tg), len(tg), _ip(out))
    assert rc == 0
    return out


def bd_strict_counts(X, targets=None):
    """relax=False, J=2: int64[m] number of pairs whose band contains the target at every t."""
    X, st, sn = _x2d(X)
    T, n = X.shape
    tg = _targets(targets, n)
    out = np.zeros(len(tg), dtype=np.int64)
    rc = lib().oracle_bd_strict_counts(_dp(X), T, n, st, sn, _lp(tg), len(tg), _ip(out))
    assert rc == 0
    return out


def bd_strict_counts_by_states(X, targets):
    """The same integers for SHORT series (T <= 8) at any n, in O(n * 2^(2T)) per target instead of O(n^2 T): numpy.

    Per (target, other curve) the state per timepoint in two bits -- above (or NaN), below (or NaN) -- is a class c; a
    pair {a, b} fails at t iff both are above or both are below there (`min <= x <= max`, _containment.py:76, with pandas'
    skipna: a NaN member of the band joins both), so it is contained at every t iff c_a & c_b == 0.  With h[c] curves
    per class the ordered contained pairs are sum over c & c' == 0 of h[c] h[c'].  Pinned against `bd_strict_counts`
    (the restatement of _functional.py:246-251 with `c // T`) in tests/test_oracle_golden.py; used where that one's
    O(n^2) per target is out of reach (10^6 points in R^3)."""
    X = np.asarray(X, dtype=np.float64)
    T, n = X.shape
    assert T <= 8
    tg = _targets(targets, n)
    out = np.zeros(len(tg), dtype=np.int64)
    NC = 1 << (2 * T)
    for k, q in enumerate(tg):
        x = X[:, q]
        if np.isnan(x).any():
            continue
        code = np.zeros(n, dtype=np.int64)
        for t in range(T):
            isn = np.isnan(X[t])
            code |= ((X[t] > x[t]) | isn).astype(np.int64) << (2 * t)
            code |= ((X[t] < x[t]) | isn).astype(np.int64) << (2 * t + 1)
        code = np.delete(code, q)
        h = np.bincount(code, minlength=NC)
        used = np.flatnonzero(h)                                    # the classes that occur (4^T of them is 65 536 at T = 8)
        hu = h[used].astype(object)                                 # Python integers: no overflow at n = 10^6
        ordered = sum(int(hu[j]) * int(hu[(used & c) == 0].sum()) for j, c in enumerate(used))
        out[k] = (ordered - int(h[0])) // 2                         # class 0 (ties everywhere) is compatible with itself
    return out


def point_in_hull(P, x, tol=DEFAULT_TOL):
    P = np.ascontiguousarray(P, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    k, d = P.shape
    rc = lib().oracle_point_in_hull(_dp(P), k, d, _dp(x), tol)
    assert rc >= 0
    return bool(rc)


def pointcloud_simplex_counts(P, targets=None, tol=DEFAULT_TOL):
    P = np.ascontiguousarray(P, dtype=np.float64)
    n, d = P.shape
    tg = _targets(targets, n)
    out = np.zeros(len(tg), dtype=np.int64)
    rc = lib().oracle_pointcloud_simplex_counts(_dp(P), n, d, _lp(tg), len(tg), tol, _ip(out))
    assert rc == 0
    return out


def multi_simplex_counts(P, targets=None, relax=True, tol=DEFAULT_TOL):
    """P: (n, T, d) curves.  int64[m] sums over (d+1)-subsets of the others."""
    P = np.ascontiguousarray(P, dtype=np.float64)
    n, T, d = P.shape
    tg = _targets(targets, n)
    out = np.zeros(len(tg), dtype=np.int64)
    rc = lib().oracle_multi_simplex_counts(_dp(P), n, T, d, _lp(tg), len(tg), int(bool(relax)), tol, _ip(out))
    assert rc == 0
    return out


def simplex_sampled(P, targets=None, relax=True, tol=DEFAULT_TOL, samples=64, seed=0):
    """Seeded subset-sampling estimator (build's own definition; see oracle.c).  P: (n, d) or (n, T, d)."""
    P = np.ascontiguousarray(P, dtype=np.float64)
    if P.ndim == 2:
        P = P[:, None, :]
    n, T, d = P.shape
    tg = _targets(targets, n)
    out = np.zeros(len(tg), dtype=np.int64)
    rc = lib().oracle_simplex_sampled(_dp(P), n, T, d, _lp(tg), len(tg), int(bool(relax)), tol, int(samples),
                                      int(seed), _ip(out))
    assert rc == 0
    return out


def l1_depth(P, targets=None):
    P = np.ascontiguousarray(P, dtype=np.float64)
    n, d = P.shape
    tg = _targets(targets, n)
    out = np.zeros(len(tg), dtype=np.float64)
    rc = lib().oracle_l1_depth(_dp(P), n, d, _lp(tg), len(tg), _dp(out))
    assert rc == 0
    return out


# ---------------------------------------------------------------------------
# float normalisers of the reference (host arithmetic, fp64)
# ---------------------------------------------------------------------------
def _binom(n, k):
    from scipy.special import binom   # the reference's normaliser (_functional.py:9,253)
    return binom(n, k)


def univariate_depths(X, targets=None, J=2, relax=False):
    """_univariate_band_depth (_functional.py:228-255): sum_j S_nj / binom(n, j), n INCLUDING the target."""
    X = np.asarray(X, dtype=np.float64)
    T, n = X.shape
    if relax:
        c = mbd_counts(X, targets, J).astype(np.float64) / T        # S_nj
    else:
        if J == 2:
            c = bd_strict_counts(X, targets).astype(np.float64)[:, None]
        else:
            c = band_enum(X, targets, J, relax=False).astype(np.float64)
    d = np.zeros(c.shape[0])
    for j in range(2, J + 1):
        d += c[:, j - 2] / _binom(n, j)
    return d


def multivariate_depths(P, targets=None, relax=False, tol=DEFAULT_TOL):
    """_simplex_depth (_functional.py:277-286): / binom(n_others, d+1)."""
    P = np.asarray(P, dtype=np.float64)
    n, T, d = P.shape
    c = multi_simplex_counts(P, targets, relax, tol).astype(np.float64)
    if relax:
        c = c / T
    return c / _binom(n - 1, d + 1)


def pointcloud_depths(P, targets=None, tol=DEFAULT_TOL):
    """_pointwisedepth simplex (_pointcloud.py:38,56): / binom(n, d+1), n INCLUDING the point."""
    P = np.asarray(P, dtype=np.float64)
    n, d = P.shape
    return pointcloud_simplex_counts(P, targets, tol).astype(np.float64) / _binom(n, d + 1)
