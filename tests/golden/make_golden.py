#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REFERENCE itself.

Runs only in the build container, where the read-only reference checkout is
mounted at /root/reference (it never travels to the GPU box).  The reference is
imported, never copied: this script holds inputs (seeded recipes) and asks the
reference for outputs.  Results are committed as small JSON fixtures next to
this script; tests read only the JSON.

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 \
        python3 tests/golden/make_golden.py [--only NAME ...] [--jobs 6]

Each fixture: {"name", "kind", "call": kwargs, "input": ..., "depths": [...],
"index": [...], "normaliser": ..., "counts": [...] (integer numerators where
the depth is an integer count over a fixed normaliser), "ref": reference
file:line the case pins, "elapsed_s"}.
"""
import argparse
import json
import math
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("STATDEPTH_REFERENCE", "/root/reference")


def _ref():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import statdepth  # noqa: F401  (the reference package)
    from statdepth import FunctionalDepth, PointcloudDepth
    from statdepth.testing import (generate_noisy_multivariate,
                                   generate_noisy_pointcloud,
                                   generate_noisy_univariate)
    return dict(FunctionalDepth=FunctionalDepth, PointcloudDepth=PointcloudDepth,
                gu=generate_noisy_univariate, gm=generate_noisy_multivariate,
                gp=generate_noisy_pointcloud)


def _pandas2_append_shim():
    """pandas >= 2 removed DataFrame.append, which two callers of the hot path still use (_pointcloud.py:118,
    homogeneity.py:173).  Inside THIS generator process only, give the frame class the pandas-1 method back (a row
    Series becomes a one-row frame labelled by its name, then pd.concat) so that the reference itself can produce the
    expected values for those callers.  Nothing of this reaches the product or the tests."""
    if hasattr(pd.DataFrame, "append"):
        return

    def append(self, other, ignore_index=False, verify_integrity=False, sort=False):
        if isinstance(other, pd.Series):
            other = other.to_frame().T
        return pd.concat([self, other], ignore_index=ignore_index, verify_integrity=verify_integrity, sort=sort)
    pd.DataFrame.append = append


def _enc(v):
    """JSON-safe float: finite floats as numbers, specials as strings."""
    v = float(v)
    if math.isnan(v):
        return "nan"
    if math.isinf(v):
        return "inf" if v > 0 else "-inf"
    return v


def _lab(i):
    return int(i) if isinstance(i, (int, np.integer)) else str(i)


def _frame_json(df):
    return {"columns": [_lab(c) for c in df.columns],
            "index": [_lab(i) for i in df.index],
            "values": [[_enc(v) for v in row] for row in df.to_numpy(dtype=float).tolist()]}


def _series_json(s):
    return {"index": [_lab(i) for i in s.index],
            "depths": [_enc(v) for v in np.asarray(s, dtype=float)]}


def _counts(depths, normaliser):
    """Integer numerators depth*normaliser, with the rounding residual."""
    raw = [float(d) * normaliser for d in depths]
    cnt = [int(round(r)) for r in raw]
    resid = max([abs(r - c) for r, c in zip(raw, cnt)] or [0.0])
    return cnt, resid


def binom(n, k):
    return math.comb(int(n), int(k))


# ----------------------------------------------------------------------------
# case builders: each returns a dict (the fixture)
# ----------------------------------------------------------------------------

def univariate_case(name, df, ref, J=2, relax=False, to_compute=None, containment="r2"):
    R = _ref()
    t0 = time.time()
    kw = dict(J=J, relax=relax, containment=containment)
    if to_compute is not None:
        kw["to_compute"] = to_compute
    s = R["FunctionalDepth"]([df], **kw)
    el = time.time() - t0
    T, n = df.shape
    out = {"name": name, "kind": "univariate", "ref": ref,
           "call": {"J": J, "relax": relax, "containment": containment, "to_compute": to_compute},
           "input": _frame_json(df), "elapsed_s": el}
    out.update(_series_json(s))
    if J == 2:
        # depth = S_n2 / C(n,2); S_n2 = (sum_t contained pairs)/T (relax) or #pairs (strict)
        norm = binom(n, 2) * (T if relax else 1)
        has_nan = bool(np.isnan(df.to_numpy(dtype=float)).any())
        cnt, resid = _counts(out["depths"], norm)
        out["normaliser"] = norm
        out["counts"] = cnt
        out["count_residual"] = resid
        out["has_nan"] = has_nan
    return out


def docs_frame():
    # docs/index.md:20-27 (6 curves x 5 timepoints) -- input data only
    return pd.DataFrame({"f_0": [1, 2, 3, 2, 1], "f_1": [2, 4, 5, 6, 2], "f_2": [3, 4, 4, 2, 1],
                         "f_3": [6, 7, 6.5, 6, 7], "f_4": [9, 9, 12, 11, 11], "f_5": [8, 8, 10, 10, 9]},
                        index=[f"x_{i}" for i in range(5)])


def build_cases():
    cases = []

    def add(fn, *a, **k):
        cases.append((fn, a, k))

    # G1 docs KAT --------------------------------------------------------
    for relax in (False, True):
        for J in (2, 3):
            add(univariate_case, f"g1_docs_J{J}_{'relax' if relax else 'strict'}", docs_frame(),
                "docs/index.md:20-41", J=J, relax=relax)
    add(univariate_case, "g1_docs_to_compute", docs_frame(), "_functional.py:70-71",
        J=2, relax=True, to_compute=["f_3", "f_0"])

    # G2 tutorial recipe -------------------------------------------------
    def g2_frame():
        return _ref()["gu"](data=[2, 3, 3.4, 4, 5, 3.1, 3, 3, 2] * 3,
                            columns=[f"f{i}" for i in range(20)], seed=42)
    for relax in (False, True):
        add(lambda name, relax=relax: univariate_case(name, g2_frame(), "tutorial/reproduce.py:29-32",
                                                      J=2, relax=relax),
            f"g2_tutorial_{'relax' if relax else 'strict'}")
    add(lambda name: univariate_case(name, g2_frame().iloc[:9, :12], "_functional.py:238-253", J=3, relax=True),
        "g2_tutorial_J3_relax_sub")

    # G7 config-1 scale (50 curves x 100 timepoints) -----------------------
    def g7_frame():
        return pd.DataFrame(np.random.default_rng(0).normal(size=(100, 50)))
    for relax in (True, False):
        add(lambda name, relax=relax: univariate_case(name, g7_frame(), "BASELINE.json configs[0]",
                                                      J=2, relax=relax),
            f"g7_config1_{'relax' if relax else 'strict'}")

    # G8 ties / duplicates / NaN ----------------------------------------------
    def g8_int(T, n, seed):
        rng = np.random.default_rng(seed)
        return pd.DataFrame(rng.integers(0, 4, size=(T, n)).astype(float),
                            columns=[f"c{i}" for i in range(n)])
    for (T, n, seed) in ((8, 5, 11), (10, 6, 12)):
        for relax in (False, True):
            add(lambda name, T=T, n=n, seed=seed, relax=relax:
                univariate_case(name, g8_int(T, n, seed), "_containment.py:76 (ties inclusive)",
                                J=2, relax=relax),
                f"g8_int_{T}x{n}_{'relax' if relax else 'strict'}")
    add(lambda name: univariate_case(name, g8_int(10, 6, 12), "_functional.py:238-253", J=3, relax=True),
        "g8_int_10x6_J3_relax")
    add(lambda name: univariate_case(name, g8_int(10, 7, 13), "_functional.py:238-253", J=4, relax=True),
        "g8_int_10x7_J4_relax")
    add(lambda name: univariate_case(name, g8_int(10, 7, 13), "_functional.py:238-253", J=3, relax=False),
        "g8_int_10x7_J3_strict")

    def g8_dup():
        rng = np.random.default_rng(21)
        df = pd.DataFrame(rng.normal(size=(12, 7)), columns=[f"d{i}" for i in range(7)])
        df["d7"] = df["d2"]          # exact duplicate curve
        df["d8"] = df["d2"]
        return df
    for relax in (False, True):
        add(lambda name, relax=relax: univariate_case(name, g8_dup(), "_containment.py:76",
                                                      J=2, relax=relax),
            f"g8_dup_{'relax' if relax else 'strict'}")

    def g8_nan():
        return pd.DataFrame({"a": [1, np.nan, 3], "b": [2, 5, np.nan], "c": [1.5, 5, 3], "d": [0, 0, 0]},
                            dtype=float)
    for relax in (True, False):
        add(lambda name, relax=relax: univariate_case(name, g8_nan(), "_containment.py:68-69 (skipna)",
                                                      J=2, relax=relax),
            f"g8_nan_{'relax' if relax else 'strict'}")

    def g8_nan2():
        rng = np.random.default_rng(31)
        a = rng.integers(0, 5, size=(9, 8)).astype(float)
        a[rng.random(a.shape) < 0.15] = np.nan
        return pd.DataFrame(a, columns=[f"n{i}" for i in range(8)])
    for relax in (True, False):
        for J in (2, 3):
            add(lambda name, relax=relax, J=J: univariate_case(name, g8_nan2(), "_containment.py:68-69 (skipna)",
                                                               J=J, relax=relax),
                f"g8_nan2_J{J}_{'relax' if relax else 'strict'}")

    def g8_inf():
        rng = np.random.default_rng(33)
        a = rng.integers(0, 4, size=(7, 7)).astype(float)
        a[0, 1] = np.inf
        a[0, 2] = np.inf
        a[3, 4] = -np.inf
        a[5, 0] = np.nan
        a[5, 6] = np.inf
        return pd.DataFrame(a, columns=[f"i{i}" for i in range(7)])
    for relax in (True, False):
        add(lambda name, relax=relax: univariate_case(name, g8_inf(), "_containment.py:68-77", J=2, relax=relax),
            f"g8_inf_{'relax' if relax else 'strict'}")

    # curve-major (column-built, F-contiguous) random walk, T not multiple of 64
    def g10():
        rng = np.random.default_rng(77)
        return pd.DataFrame({f"w{i}": rng.normal(size=37).cumsum() for i in range(23)})
    for relax in (True, False):
        add(lambda name, relax=relax: univariate_case(name, g10(), "_functional.py:198-255", J=2, relax=relax),
            f"g10_walk_37x23_{'relax' if relax else 'strict'}")

    # G12: what the strict path's complement matching treats specially -- curves sharing their first value (a constant
    # timepoint), mirrored curves (complementary masks on purpose), a curve that touches another once (sparse tie), and
    # more than 32 timepoints (two mask words) -- from the reference itself
    def g12(kind):
        rng = np.random.default_rng(212)
        T, n = 40, 16
        X = np.sort(rng.normal(size=n))[None, :] * 1.5 + rng.normal(size=(T, n)).cumsum(axis=0) * 0.15
        if kind == "common_start":
            X = X - X[0:1, :]
        elif kind == "mirrored":
            X[:, n // 2:] = -X[:, : n // 2]
            X[:, 0] = 0.0
        elif kind == "touch":
            X[17, 3] = X[17, 9]
            X[29, 5] = X[29, 6]
        return pd.DataFrame(X, columns=[f"c{i}" for i in range(n)])
    for kind in ("common_start", "mirrored", "touch"):
        add(lambda name, kind=kind: univariate_case(name, g12(kind), "_functional.py:198-255", J=2, relax=False),
            f"g12_{kind}_strict")

    # G9 K-sampled ----------------------------------------------------------
    def g9(name):
        R = _ref()
        df = R["gu"](n=20, seed=1)
        np.random.seed(7)
        t0 = time.time()
        s = R["FunctionalDepth"]([df], K=5, relax=True)
        out = {"name": name, "kind": "univariate_sampled", "ref": "_functional.py:153-186",
               "call": {"K": 5, "J": 2, "relax": True, "np_random_seed": 7},
               "input": _frame_json(df), "elapsed_s": time.time() - t0}
        out.update(_series_json(s))
        return out
    add(g9, "g9_ksampled")

    def g9s(name):
        # the same estimator with the reference's default relax=False: smooth, well separated curves so that whole
        # blocks contain their target at every timepoint
        R = _ref()
        rng = np.random.default_rng(19)
        t = np.linspace(0, 1, 25)
        df = pd.DataFrame({f"c{i}": lvl + 0.15 * np.sin(2 * np.pi * (t + ph))
                           for i, (lvl, ph) in enumerate(zip(rng.normal(size=24) * 2.0, rng.random(24)))})
        np.random.seed(11)
        t0 = time.time()
        s = R["FunctionalDepth"]([df], K=3, relax=False)
        out = {"name": name, "kind": "univariate_sampled", "ref": "_functional.py:153-186",
               "call": {"K": 3, "J": 2, "relax": False, "np_random_seed": 11},
               "input": _frame_json(df), "elapsed_s": time.time() - t0}
        out.update(_series_json(s))
        return out
    add(g9s, "g9_ksampled_strict")

    # G3 / G4 multivariate simplex -------------------------------------------
    def multi_case(name, frames, ref, relax, to_compute=None):
        R = _ref()
        t0 = time.time()
        kw = dict(containment="simplex", relax=relax)
        if to_compute is not None:
            kw["to_compute"] = to_compute
        s = R["FunctionalDepth"](frames, **kw)
        n = len(frames)
        T, d = frames[0].shape
        out = {"name": name, "kind": "multivariate", "ref": ref,
               "call": {"containment": "simplex", "relax": relax, "to_compute": to_compute},
               "input": [_frame_json(f) for f in frames], "elapsed_s": time.time() - t0}
        out.update(_series_json(s))
        norm = binom(n - 1, d + 1) * (T if relax else 1)
        cnt, resid = _counts(out["depths"], norm)
        out.update(normaliser=norm, counts=cnt, count_residual=resid)
        return out

    for relax in (True, False):
        add(lambda name, relax=relax: multi_case(
            name, _ref()["gm"](columns=list("ABC"), num_curves=10, seed=42),
            "tutorial/reproduce.py:48-49 (degenerate simplices)", relax),
            f"g3_multi_degenerate_{'relax' if relax else 'strict'}")

    def g4_frames(seed=2024, n=6, T=4, d=2):
        rng = np.random.default_rng(seed)
        return [pd.DataFrame(rng.normal(size=(T, d)), columns=list("xyzw")[:d]) for _ in range(n)]
    for relax in (True, False):
        add(lambda name, relax=relax: multi_case(name, g4_frames(), "_functional.py:257-286", relax),
            f"g4_multi_nondegenerate_{'relax' if relax else 'strict'}")
    add(lambda name: multi_case(name, g4_frames(seed=2025, n=8, T=5, d=2), "_functional.py:86-89", True,
                                to_compute=[3, 0, 7]),
        "g4_multi_to_compute")
    add(lambda name: multi_case(name, g4_frames(seed=2026, n=7, T=3, d=3), "_functional.py:257-286", True),
        "g4_multi_d3")
    # curves that stay close together so that some strict containments are 1
    def g4_tight():
        rng = np.random.default_rng(2027)
        base = [pd.DataFrame(rng.normal(size=(1, 2)).repeat(3, axis=0) * 3.0) for _ in range(6)]
        cen = pd.DataFrame(np.zeros((3, 2)) + rng.normal(size=(3, 2)) * 0.01)
        return base + [cen]
    for relax in (True, False):
        add(lambda name, relax=relax: multi_case(name, g4_tight(), "_containment.py:136", relax),
            f"g4_multi_tight_{'relax' if relax else 'strict'}")

    # G5 pointcloud simplex ----------------------------------------------------
    def pc_case(name, df, ref, containment="simplex", to_compute=None):
        R = _ref()
        t0 = time.time()
        kw = dict(containment=containment)
        if to_compute is not None:
            kw["to_compute"] = to_compute
        s = R["PointcloudDepth"](df, **kw)
        n, d = df.shape
        out = {"name": name, "kind": "pointcloud", "ref": ref,
               "call": {"containment": containment, "to_compute": to_compute},
               "input": _frame_json(df), "elapsed_s": time.time() - t0}
        out.update(_series_json(s))
        if containment == "simplex":
            norm = binom(n, d + 1)
            cnt, resid = _counts(out["depths"], norm)
            out.update(normaliser=norm, counts=cnt, count_residual=resid)
        return out

    add(lambda name: pc_case(name, _ref()["gp"](n=30, d=2, seed=42), "_pointcloud.py:44-56"), "g5_pc_n30_d2")
    add(lambda name: pc_case(name, _ref()["gp"](n=12, d=2, seed=5), "_pointcloud.py:44-56"), "g5_pc_n12_d2")
    add(lambda name: pc_case(name, _ref()["gp"](n=10, d=3, seed=6), "_pointcloud.py:44-56"), "g5_pc_n10_d3")
    add(lambda name: pc_case(name, _ref()["gp"](n=12, d=1, seed=8), "_pointcloud.py:44-56"), "g5_pc_n12_d1")
    add(lambda name: pc_case(name, _ref()["gp"](n=12, d=2, seed=5), "_pointcloud.py:41-42",
                             to_compute=[7, 2, 11]), "g5_pc_to_compute")

    def pc_grid():
        # integer lattice: many collinear triples and points on edges (degenerate + boundary)
        pts = [(x, y) for x in range(3) for y in range(3)] + [(1, 1), (0, 0)]
        return pd.DataFrame(np.array(pts, dtype=float), index=[f"p{i}" for i in range(len(pts))])
    add(lambda name: pc_case(name, pc_grid(), "_containment.py:161-176 (degenerate/boundary)"), "g5_pc_grid_d2")

    # G6 l1 ------------------------------------------------------------------
    def docs_pc():
        return pd.DataFrame([[0.873179, 0.828111], [0.368512, 0.024619], [0.927522, 0.348593],
                             [0.481917, 0.748796], [0.980515, 0.954392]])
    add(lambda name: pc_case(name, docs_pc(), "docs/index.md:98-112", containment="l1"), "g6_l1_docs")
    add(lambda name: pc_case(name, _ref()["gp"](n=40, d=3, seed=9), "_pointcloud.py:125-150", containment="l1"),
        "g6_l1_n40_d3")
    add(lambda name: pc_case(name, _ref()["gp"](n=15, d=5, seed=10), "_pointcloud.py:125-150", containment="l1",
                             to_compute=[3, 14, 0]), "g6_l1_to_compute")

    # P4: L-infinity / box containment expressed with reference semantics -------
    # FunctionalDepth([df.T]) -- SURVEY.md 8(a) row P4
    def p4_case(name, relax):
        df = _ref()["gp"](n=14, d=3, seed=15)
        out = univariate_case(name, df.T, "SURVEY 8(a) P4: FunctionalDepth([df.T])", J=2, relax=relax)
        out["kind"] = "pointcloud_linf"
        out["points"] = _frame_json(df)
        return out
    for relax in (True, False):
        add(lambda name, relax=relax: p4_case(name, relax), f"p4_linf_{'relax' if relax else 'strict'}")

    # homogeneity (caller of the hot path) -------------------------
    def homog_case(name, method):
        if REF not in sys.path:
            sys.path.insert(0, REF)
        from statdepth.homogeneity import FunctionalHomogeneity
        rng = np.random.default_rng(50)
        F = pd.DataFrame(rng.normal(size=(10, 9)).cumsum(axis=0), columns=[f"F{i}" for i in range(9)])
        G = pd.DataFrame(rng.normal(size=(10, 7)).cumsum(axis=0) + 0.5, columns=[f"G{i}" for i in range(7)])
        t0 = time.time()
        h = FunctionalHomogeneity([F.copy()], [G.copy()], method=method, relax=True, quiet=True)
        val = h.homogeneity()
        val = np.asarray(val, dtype=float).ravel()
        return {"name": name, "kind": "homogeneity", "ref": "homogeneity.py:65-153",
                "call": {"method": method, "relax": True, "J": 2},
                "input": {"F": _frame_json(F), "G": _frame_json(G)},
                "value": [_enc(v) for v in val], "elapsed_s": time.time() - t0}
    for m in ("p1", "p2", "p3"):
        add(lambda name, m=m: homog_case(name, m), f"h_{m}")

    # the same with the reference's default relax=False (the strict depth of every g inside F u {g}); banded samples so
    # that strict depths are not all zero
    def homog_strict_case(name, method):
        if REF not in sys.path:
            sys.path.insert(0, REF)
        from statdepth.homogeneity import FunctionalHomogeneity
        rng = np.random.default_rng(53)
        F = pd.DataFrame(np.sort(rng.normal(size=11))[None, :] * 2.0 + rng.normal(size=(12, 11)) * 0.2,
                         columns=[f"F{i}" for i in range(11)])
        G = pd.DataFrame(np.sort(rng.normal(size=8))[None, :] * 1.5 + rng.normal(size=(12, 8)) * 0.2 + 0.1,
                         columns=[f"G{i}" for i in range(8)])
        t0 = time.time()
        val = FunctionalHomogeneity([F.copy()], [G.copy()], method=method, relax=False, quiet=True).homogeneity()
        val = np.asarray(val, dtype=float).ravel()
        return {"name": name, "kind": "homogeneity", "ref": "homogeneity.py:65-153",
                "call": {"method": method, "relax": False, "J": 2},
                "input": {"F": _frame_json(F), "G": _frame_json(G)},
                "value": [_enc(v) for v in val], "elapsed_s": time.time() - t0}
    for m in ("p1", "p3"):
        add(lambda name, m=m: homog_strict_case(name, m), f"h_{m}_strict")

    # the function forms P1_homogeneity / P2_homogeneity (homogeneity.py:214-306).  NB the reference's P1 leaves its
    # 'G_deepest' column in the caller's F, and P2 then takes F's deepest depth WITH that column present.
    def homog_fn_case(name, which, relax):
        if REF not in sys.path:
            sys.path.insert(0, REF)
        from statdepth.homogeneity import P1_homogeneity, P2_homogeneity
        rng = np.random.default_rng(51)
        F = pd.DataFrame(rng.normal(size=(9, 8)).cumsum(axis=0), columns=[f"F{i}" for i in range(8)])
        G = pd.DataFrame(rng.normal(size=(9, 6)).cumsum(axis=0) + 0.4, columns=[f"G{i}" for i in range(6)])
        t0 = time.time()
        fn = P1_homogeneity if which == "P1" else P2_homogeneity
        val = fn(F.copy(), G.copy(), relax=relax, quiet=True)
        return {"name": name, "kind": "homogeneity_fn", "ref": "homogeneity.py:214-306",
                "call": {"fn": which, "relax": relax, "J": 2},
                "input": {"F": _frame_json(F), "G": _frame_json(G)},
                "value": [_enc(float(val))], "elapsed_s": time.time() - t0}
    for which in ("P1", "P2"):
        for relax in (True, False):
            add(lambda name, which=which, relax=relax: homog_fn_case(name, which, relax),
                f"h_{which}fn_{'relax' if relax else 'strict'}")

    # P3 with OVERLAPPING labels (the default RangeIndex case): the reference's loop overwrites F's own column with
    # g and then drops it (homogeneity.py:125-128), so F shrinks as the loop runs.  Recorded for documentation: the
    # build deliberately evaluates every g inside the intact F u {g} instead (tests/test_hip_parity.py).
    def homog_overlap_case(name):
        if REF not in sys.path:
            sys.path.insert(0, REF)
        from statdepth.homogeneity import FunctionalHomogeneity
        rng = np.random.default_rng(52)
        F = pd.DataFrame(rng.normal(size=(8, 7)).cumsum(axis=0))
        G = pd.DataFrame(rng.normal(size=(8, 5)).cumsum(axis=0) + 0.3)
        t0 = time.time()
        val = FunctionalHomogeneity([F.copy()], [G.copy()], method="p3", relax=True, quiet=True).homogeneity()
        return {"name": name, "kind": "homogeneity_overlap", "ref": "homogeneity.py:125-133",
                "call": {"method": "p3", "relax": True, "J": 2},
                "input": {"F": _frame_json(F), "G": _frame_json(G)},
                "reference_value": [_enc(v) for v in np.asarray(val, dtype=float).ravel()],
                "elapsed_s": time.time() - t0}
    add(homog_overlap_case, "h_p3_overlap")

    # duplicate column labels: _subsequences collapses label tuples through a set (_helper.py:32) and `.loc` with a
    # duplicated label returns every column carrying it.  What the reference does, recorded per call.
    def duplabel_case(name):
        ref = _ref()
        rng = np.random.default_rng(3)
        X = np.round(rng.normal(size=(6, 5)).cumsum(axis=0), 1)
        df = pd.DataFrame(X, columns=["a", "b", "b", "c", "d"])
        t0 = time.time()
        calls = []
        for relax in (True, False):
            for tc in (["a"], ["c", "d"], ["b"], None):
                try:
                    r = ref["FunctionalDepth"]([df.copy()], to_compute=tc, relax=relax)
                    calls.append({"relax": relax, "to_compute": tc, "depths": [_enc(v) for v in r.to_numpy()]})
                except Exception as e:          # noqa: BLE001 -- the exception type is the recorded behaviour
                    calls.append({"relax": relax, "to_compute": tc, "raises": type(e).__name__})
        return {"name": name, "kind": "duplabels", "ref": "_helper.py:32, _functional.py:232-248",
                "input": _frame_json(df), "calls": calls, "elapsed_s": time.time() - t0}
    add(duplabel_case, "g11_duplabels")

    # P2: the K-sampled point-cloud depth (_samplepointwisedepth, _pointcloud.py:68-123) -- runs here only with the
    # DataFrame.append shim; the draws come from the global numpy RNG (data.sample), seeded per case
    def pc_sampled_case(name, containment, to_compute, K, seed):
        R = _ref()
        _pandas2_append_shim()
        rng = np.random.default_rng(23)
        n, d = 26, 2
        df = pd.DataFrame(rng.normal(size=(n, d)), index=[f"p{i}" for i in range(n)], columns=["x", "y"])
        np.random.seed(seed)
        t0 = time.time()
        s = R["PointcloudDepth"](df, to_compute=to_compute, K=K, containment=containment)
        out = {"name": name, "kind": "pointcloud_sampled", "ref": "_pointcloud.py:68-123",
               "call": {"containment": containment, "to_compute": to_compute, "K": K, "np_random_seed": seed},
               "input": _frame_json(df), "elapsed_s": time.time() - t0}
        out.update(_series_json(s))
        return out
    add(lambda name: pc_sampled_case(name, "simplex", ["p3", "p0", "p25", "p11"], 3, 99), "p2_ksampled_simplex")
    add(lambda name: pc_sampled_case(name, "l1", ["p3", "p0", "p25", "p11"], 3, 99), "p2_ksampled_l1")
    add(lambda name: pc_sampled_case(name, "simplex", None, 2, 5), "p2_ksampled_simplex_all")
    add(lambda name: pc_sampled_case(name, "l1", None, 4, 6), "p2_ksampled_l1_all")

    # point-cloud homogeneity (_pointcloudhomogeneity, homogeneity.py:155-200; F.append at :173): same shim.  The
    # reference demands len(F) == len(G) (:204-205).  'distinct': F and G carry labels of their own; 'shared': the
    # default RangeIndex on both, where the reference's P3 loop overwrites and drops F's own rows (:183-186) -- kept
    # as the reference's value for the record.  P4 raises TypeError in the reference (tuple - tuple, :194-195).
    def pc_homog_case(name, labels, method, containment):
        if REF not in sys.path:
            sys.path.insert(0, REF)
        _pandas2_append_shim()
        from statdepth.homogeneity import PointcloudHomogeneity
        rng = np.random.default_rng(31)
        F = pd.DataFrame(rng.normal(size=(13, 2)))
        G = pd.DataFrame(rng.normal(size=(13, 2)) * 0.8 + 0.2)
        if labels == "distinct":
            F.index = [f"f{i}" for i in range(13)]
            G.index = [f"g{i}" for i in range(13)]
        t0 = time.time()
        out = {"name": name, "kind": "pointcloud_homogeneity", "ref": "homogeneity.py:155-200",
               "call": {"method": method, "containment": containment, "labels": labels},
               "input": {"F": _frame_json(F), "G": _frame_json(G)}}
        try:
            h = PointcloudHomogeneity(F.copy(), G.copy(), method=method, containment=containment)
            out["value"] = [_enc(v) for v in np.asarray(h.homogeneity(), dtype=float).ravel()]
        except Exception as e:                  # noqa: BLE001 -- the exception type is the recorded behaviour
            out["raises"] = type(e).__name__
        out["elapsed_s"] = time.time() - t0
        return out
    for lab in ("distinct", "shared"):
        for m in ("p1", "p2", "p3", "p4"):
            for c in ("simplex", "l1"):
                add(lambda name, lab=lab, m=m, c=c: pc_homog_case(name, lab, m, c), f"hpc_{lab}_{m}_{c}")

    return cases


def _run(idx):
    cases = build_cases()
    fn, a, k = cases[idx]
    # first positional arg is the name in every builder
    out = fn(*a, **k)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--jobs", type=int, default=6)
    args = ap.parse_args()
    cases = build_cases()
    names = [a[0] for (_, a, _) in cases]
    assert len(set(names)) == len(names), "duplicate fixture names"
    todo = [i for i, nm in enumerate(names) if args.only is None or nm in args.only]
    with ProcessPoolExecutor(max_workers=args.jobs) as ex:
        for out in ex.map(_run, todo):
            path = os.path.join(HERE, out["name"] + ".json")
            with open(path, "w") as f:
                json.dump(out, f, indent=None, separators=(",", ":"), allow_nan=False)
                f.write("\n")
            print(f"wrote {out['name']}: {out['elapsed_s']:.1f}s", flush=True)


if __name__ == "__main__":
    main()
