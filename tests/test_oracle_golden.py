"""Pin the CPU oracle against golden vectors produced by the reference itself.

The fixtures under tests/golden/*.json were written by tests/golden/make_golden.py,
which imports the reference (in the build container only) and records its outputs.
Integer containment counts must match bit-exact; normalised depths within 1e-12.
"""
import numpy as np
import pytest

from conftest import (assert_depths_close, depths_of, frame_values, golden_names,
                      load_golden)

TOL = 1e-12


def _targets(fx, labels):
    tc = fx["call"].get("to_compute")
    if tc is None:
        return None
    return [labels.index(t) for t in tc]


@pytest.mark.parametrize("name", golden_names(kind="univariate") + golden_names(kind="pointcloud_linf"))
def test_univariate_depths(oracle, name):
    fx = load_golden(name)
    X = frame_values(fx["input"])                      # (T, n): rows = timepoints
    J, relax = fx["call"]["J"], fx["call"]["relax"]
    tg = _targets(fx, fx["input"]["columns"])
    got = oracle.univariate_depths(X, tg, J=J, relax=relax)
    assert_depths_close(got, depths_of(fx), TOL)
    # integer numerators bit-exact where the fixture carries them (J == 2)
    if "counts" in fx and fx["count_residual"] < 1e-6:
        if relax:
            cnt = oracle.mbd_counts(X, tg, 2)[:, 0]
        else:
            cnt = oracle.bd_strict_counts(X, tg)
        assert cnt.tolist() == fx["counts"]


@pytest.mark.parametrize("name", [n for n in golden_names(kind="univariate")
                                  if load_golden(n)["call"]["relax"]
                                  and len(load_golden(n)["input"]["columns"]) <= 25])
def test_closed_form_equals_literal_enumeration(oracle, name):
    """C(v,k)-C(A,k)-C(B,k) closed form (with NaN weights) == literal subset enumeration."""
    fx = load_golden(name)
    X = frame_values(fx["input"])
    J = max(fx["call"]["J"], 3)
    if J >= X.shape[0]:
        J = fx["call"]["J"]
    a = oracle.mbd_counts(X, None, J)
    b = oracle.band_enum(X, None, J, relax=True)
    assert (a == b).all()


@pytest.mark.parametrize("name", [n for n in golden_names(kind="univariate") if load_golden(n)["call"]["relax"]])
def test_rank_sort_formulation_equals_closed_form(oracle, name):
    """oracle_mbd_counts_ranksort (per-timepoint sort; bench.py's like-for-like CPU baseline of the rank kernels) gives
    the integers of the pairwise closed form on every relax fixture, ties / NaN / +-inf included."""
    fx = load_golden(name)
    X = frame_values(fx["input"])
    for J in (2, 3):
        if J < X.shape[1]:
            assert (oracle.mbd_counts_ranksort(X, J) == oracle.mbd_counts(X, None, J)).all()
            assert (oracle.mbd_counts_ranksort(np.asfortranarray(X), J) == oracle.mbd_counts(X, None, J)).all()


@pytest.mark.parametrize("name", [n for n in golden_names(kind="univariate")
                                  if not load_golden(n)["call"]["relax"]
                                  and load_golden(n)["call"]["J"] == 2
                                  and len(load_golden(n)["input"]["columns"]) <= 25])
def test_strict_bitmask_equals_literal_enumeration(oracle, name):
    fx = load_golden(name)
    X = frame_values(fx["input"])
    a = oracle.bd_strict_counts(X, None)
    b = oracle.band_enum(X, None, 2, relax=False)[:, 0]
    assert (a == b).all()


def test_both_layouts_agree(oracle):
    rng = np.random.default_rng(5)
    X = rng.integers(0, 6, size=(19, 13)).astype(float)
    XF = np.asfortranarray(X)
    assert (oracle.mbd_counts(X, None, 3) == oracle.mbd_counts(XF, None, 3)).all()
    assert (oracle.bd_strict_counts(X) == oracle.bd_strict_counts(XF)).all()
    assert (oracle.above_below(X) == oracle.above_below(XF)).all()


def test_numpy_restatement_agrees(oracle):
    """Second, independently written restatement (oracle_np) on a tie-heavy case."""
    from oracle import oracle_np
    rng = np.random.default_rng(8)
    X = rng.integers(0, 4, size=(7, 8)).astype(float)
    X[2, 3] = np.nan
    for relax in (True, False):
        for J in (2, 3):
            want = [oracle_np.univariate_band_depth(X, i, J, relax) for i in range(X.shape[1])]
            got = oracle.univariate_depths(X, None, J=J, relax=relax)
            assert_depths_close(got, want, TOL)


@pytest.mark.parametrize("name", golden_names(kind="multivariate"))
def test_multivariate_simplex(oracle, name):
    fx = load_golden(name)
    P = np.stack([frame_values(f) for f in fx["input"]])        # (n, T, d)
    relax = fx["call"]["relax"]
    tg = fx["call"].get("to_compute")
    got = oracle.multivariate_depths(P, tg, relax=relax)
    assert_depths_close(got, depths_of(fx), TOL)
    assert fx["count_residual"] < 1e-6
    assert oracle.multi_simplex_counts(P, tg, relax).tolist() == fx["counts"]


@pytest.mark.parametrize("name", golden_names(kind="pointcloud"))
def test_pointcloud(oracle, name):
    fx = load_golden(name)
    P = frame_values(fx["input"])
    tg = _targets(fx, fx["input"]["index"])
    if fx["call"]["containment"] == "simplex":
        got = oracle.pointcloud_depths(P, tg)
        assert_depths_close(got, depths_of(fx), TOL)
        assert oracle.pointcloud_simplex_counts(P, tg).tolist() == fx["counts"]
    else:
        got = oracle.l1_depth(P, tg)
        assert_depths_close(got, depths_of(fx), TOL)


def _sampled_pointcloud_restatement(oracle, df, labels, K, containment):
    """_samplepointwisedepth (_pointcloud.py:105-121) with the oracle as the exact depth inside each sample: per point
    ss = n // K draws of data.sample(n=ss) from the global numpy RNG, the point appended when the draw missed it."""
    import pandas as pd
    n = len(df)
    ss = n // K
    out = []
    for lab in labels:
        vals = []
        for _ in range(ss):
            sdata = df.sample(n=ss, axis=0)
            if lab not in sdata.index:
                sdata = pd.concat([sdata, df.loc[[lab], :]])
            pos = list(sdata.index).index(lab)
            fn = oracle.pointcloud_depths if containment == "simplex" else oracle.l1_depth
            vals.append(fn(sdata.to_numpy(), [pos])[0])
        out.append(np.mean(vals))
    return np.array(out)


@pytest.mark.parametrize("name", golden_names(kind="pointcloud_sampled"))
def test_sampled_pointcloud_depth_vs_reference(oracle, name):
    """P2 pinned to the reference itself (fixtures written by running _samplepointwisedepth under a DataFrame.append shim
    in the generator): the oracle-based restatement of the draw loop reproduces the reference's values."""
    from conftest import frame_df
    fx = load_golden(name)
    df = frame_df(fx["input"])
    labels = fx["call"]["to_compute"] or list(df.index)
    np.random.seed(fx["call"]["np_random_seed"])
    got = _sampled_pointcloud_restatement(oracle, df, labels, fx["call"]["K"], fx["call"]["containment"])
    assert fx["index"] == labels
    assert_depths_close(got, depths_of(fx), TOL)


def pointcloud_homogeneity_restatement(oracle, Fx, Gx, method, containment):
    """_pointcloudhomogeneity (homogeneity.py:155-200) on label-disjoint samples, from oracle depths: `median()` of a
    depth result is its deepest value; every g is evaluated inside F u {g} (n_F + 1 points)."""
    def depths(P):
        return oracle.pointcloud_depths(P) if containment == "simplex" else oracle.l1_depth(P)

    def inside(host, pt):
        Hp = np.concatenate([host, pt[None, :]])
        return (oracle.pointcloud_depths(Hp, [len(host)]) if containment == "simplex" else oracle.l1_depth(Hp, [len(host)]))[0]

    dF, dG = depths(Fx), depths(Gx)
    g_in_F = inside(Fx, Gx[np.argmax(dG)])
    if method == "p1":
        return g_in_F / dF.max()
    if method == "p2":
        return 1 - abs(g_in_F - dF.max())
    p3 = max(inside(Fx, g) for g in Gx) / dG.max()
    if method == "p3":
        return p3
    return abs(p3 - inside(Fx, Fx[np.argmax(dF)]) / dF.max()) * abs(p3 - inside(Gx, Gx[np.argmax(dG)]) / dG.max())


@pytest.mark.parametrize("name", golden_names(kind="pointcloud_homogeneity"))
def test_pointcloud_homogeneity_vs_reference(oracle, name):
    """Point-cloud homogeneity pinned to the reference itself (same shim).  Label-disjoint samples: P1-P3 equal the
    restatement.  Shared labels (default RangeIndex): P1 / P2 still do; P3 is where the reference overwrites and drops
    F's own rows (homogeneity.py:183-186) -- its value is on record and is NOT the definition's.  P4 raises TypeError in
    the reference (tuple - tuple, :194-195)."""
    fx = load_golden(name)
    c = fx["call"]
    if c["method"] == "p4":
        assert fx.get("raises") == "TypeError"
        return
    Fx, Gx = frame_values(fx["input"]["F"]), frame_values(fx["input"]["G"])
    want = float(fx["value"][0])
    got = pointcloud_homogeneity_restatement(oracle, Fx, Gx, c["method"], c["containment"])
    if c["labels"] == "shared" and c["method"] == "p3":
        assert abs(got - want) > 1e-6
    else:
        assert abs(got - want) <= TOL * max(1.0, abs(want))


def test_hull_restatement_vs_lp_call(oracle):
    """Geometric restatement == the reference's third-party LP call on random and degenerate sets."""
    from oracle import oracle_np
    rng = np.random.default_rng(17)
    n_checked = 0
    for d in (1, 2, 3, 4):
        for _ in range(40):
            P = rng.normal(size=(d + 1, d))
            x = rng.normal(size=d) * 0.4
            assert oracle.point_in_hull(P, x) == oracle_np.is_in_simplex_lp(P, x)
            n_checked += 1
    # degenerate: all points on one ray (the reference's own multivariate generator shape)
    for _ in range(40):
        base = rng.random(3)
        r = rng.random(4)
        P = np.outer(r, base)
        x = rng.random() * base
        assert oracle.point_in_hull(P, x) == oracle_np.is_in_simplex_lp(P, x)
    # boundary: lattice points on edges / vertices are contained (closed simplex)
    tri = np.array([[0.0, 0.0], [2.0, 0.0], [0.0, 2.0]])
    for x in ([1.0, 0.0], [1.0, 1.0], [0.0, 0.0], [2.0, 0.0], [0.5, 0.5]):
        assert oracle.point_in_hull(tri, np.array(x)) is True
        assert oracle_np.is_in_simplex_lp(tri, x) is True
    for x in ([1.0, 1.0 + 1e-5], [-1e-5, 0.5], [2.1, 0.0]):
        assert oracle.point_in_hull(tri, np.array(x)) is False
        assert oracle_np.is_in_simplex_lp(tri, x) is False


def test_l1_matches_numpy_restatement(oracle):
    from oracle import oracle_np
    rng = np.random.default_rng(3)
    P = rng.normal(size=(17, 4))
    want = [oracle_np.l1_depth(P, i) for i in range(17)]
    assert_depths_close(oracle.l1_depth(P), want, TOL)
    # coincident points give NaN in the reference (0/0, _pointcloud.py:146)
    P[5] = P[2]
    got = oracle.l1_depth(P)
    assert np.isnan(got[5]) and np.isnan(got[2]) and np.isnan(got).sum() == 2


def test_componentwise_band_enum_reduces_to_univariate(oracle):
    """'r2_enum' (_containment.py:83-103, unimplemented upstream) restated literally: with ONE feature it is the
    univariate enumeration the fixtures pin, for both relax modes, NaN included."""
    for name in ("g1_docs_J3_relax", "g8_nan2_J3_relax", "g8_int_8x5_relax"):
        X = frame_values(load_golden(name)["input"])
        P = np.ascontiguousarray(X.T[:, :, None])
        for relax in (True, False):
            assert (oracle.multi_band_enum(P, None, 3, relax) == oracle.band_enum(X, None, 3, relax)).all()
    # two features: contained iff contained in both -> never more than either feature alone
    rng = np.random.default_rng(5)
    P = np.round(rng.normal(size=(9, 6, 2)).cumsum(axis=1), 1)
    both = oracle.multi_band_enum(P, None, 2, True)[:, 0]
    for f in range(2):
        alone = oracle.band_enum(np.ascontiguousarray(P[:, :, f].T), None, 2, True)[:, 0]
        assert (both <= alone).all()


def test_oracle_is_sanitizer_clean():
    """The checker itself under AddressSanitizer + UBSan on the CPU (`make -C oracle sanitize`): every entry point driven
    on small inputs with ties, NaN and infinities; the GPU pool offers no sanitizer, so this is where memory errors in
    the test infrastructure would show."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-s", "-C", os.path.join(root, "oracle"), "sanitize"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "selftest ok" in r.stdout and "ERROR" not in r.stderr


def test_strict_counts_by_states_equal_the_enumeration():
    """oracle.bd_strict_counts_by_states (short series, any n) against oracle.bd_strict_counts (pair enumeration)."""
    import oracle
    rng = np.random.default_rng(3)
    for T, n in [(1, 7), (2, 40), (3, 150), (4, 60)]:
        for variant in range(3):
            X = rng.normal(size=(T, n))
            if variant == 1:
                X = np.round(X * 2)
            if variant == 2:
                X[rng.integers(0, T), 3] = np.nan
                X[:, 5] = X[:, 6]
            tg = np.arange(n)
            assert (oracle.bd_strict_counts_by_states(X, tg) == oracle.bd_strict_counts(X, tg)).all(), (T, n, variant)
