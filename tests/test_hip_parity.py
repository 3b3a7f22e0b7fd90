"""GPU parity: the HIP path (through the C ABI) against the golden vectors and the oracle.

Integer containment counts: bit-exact.  Normalised depths: within 1e-12 (the tolerance
BASELINE.json's north_star states).  Everything here calls the product API or
statdepth_amd.engine, i.e. libstatdepth_hip.so; the oracle is only the checker.
"""
import numpy as np
import pandas as pd
import pytest

from conftest import (assert_depths_close, depths_of, frame_df, frame_values, golden_names,
                      load_golden)

pytestmark = pytest.mark.gpu
TOL = 1e-12


@pytest.fixture(scope="module")
def eng():
    from statdepth_amd import engine
    return engine


# ---------------------------------------------------------------- golden, through the public API
@pytest.mark.parametrize("algo", ["pairwise", "rank"])
@pytest.mark.parametrize("name", golden_names(kind="univariate") + golden_names(kind="pointcloud_linf"))
def test_golden_univariate_api(name, algo):
    from statdepth_amd import FunctionalDepth
    fx = load_golden(name)
    call = fx["call"]
    if not call["relax"] and algo == "rank":
        pytest.skip("algo only selects the relax=True kernel")
    df = frame_df(fx["input"])
    got = FunctionalDepth([df], to_compute=call["to_compute"], J=call["J"], relax=call["relax"],
                          containment=call["containment"], algo=algo)
    assert isinstance(got, pd.Series)
    assert list(got.index) == fx["index"]
    assert_depths_close(got.to_numpy(), depths_of(fx), TOL)


@pytest.mark.parametrize("name", [n for n in golden_names(kind="univariate") if "counts" in load_golden(n)])
def test_golden_univariate_counts_bit_exact(eng, name):
    fx = load_golden(name)
    if fx["count_residual"] >= 1e-6:
        pytest.skip("fixture numerators not integral")
    X = frame_values(fx["input"])
    tc = fx["call"]["to_compute"]
    tg = None if tc is None else [fx["input"]["columns"].index(c) for c in tc]
    if fx["call"]["relax"]:
        for algo in ("pairwise", "rank"):
            assert eng.mbd_counts(X, tg, 2, algo=algo)[:, 0].tolist() == fx["counts"]
    else:
        assert eng.bd_strict_counts(X, tg, 2)[:, 0].tolist() == fx["counts"]


@pytest.mark.parametrize("name", golden_names(kind="multivariate"))
def test_golden_multivariate_api(eng, name):
    from statdepth_amd import FunctionalDepth
    fx = load_golden(name)
    frames = [frame_df(f) for f in fx["input"]]
    got = FunctionalDepth(frames, to_compute=fx["call"]["to_compute"], containment="simplex",
                          relax=fx["call"]["relax"])
    assert list(got.index) == fx["index"]
    assert_depths_close(got.to_numpy(), depths_of(fx), TOL)
    P = np.stack([frame_values(f) for f in fx["input"]])
    assert eng.multi_simplex_counts(P, fx["call"]["to_compute"], relax=fx["call"]["relax"]).tolist() == fx["counts"]


@pytest.mark.parametrize("name", golden_names(kind="pointcloud"))
def test_golden_pointcloud_api(eng, name):
    from statdepth_amd import PointcloudDepth
    fx = load_golden(name)
    df = frame_df(fx["input"])
    got = PointcloudDepth(df, to_compute=fx["call"]["to_compute"], containment=fx["call"]["containment"])
    assert list(got.index) == fx["index"]
    assert_depths_close(got.to_numpy(), depths_of(fx), TOL)
    if fx["call"]["containment"] == "simplex":
        tc = fx["call"]["to_compute"]
        tg = None if tc is None else [fx["input"]["index"].index(c) for c in tc]
        assert eng.pointcloud_simplex_counts(frame_values(fx["input"]), tg).tolist() == fx["counts"]


def test_golden_ksampled_reproduces_reference_blocks():
    """np.random.seed pins the reference's block draws (_functional.py:176): same depths."""
    from statdepth_amd import FunctionalDepth
    fx = load_golden("g9_ksampled")
    df = frame_df(fx["input"])
    np.random.seed(fx["call"]["np_random_seed"])
    got = FunctionalDepth([df], K=fx["call"]["K"], relax=True)
    assert_depths_close(got.to_numpy(), depths_of(fx), TOL)


@pytest.mark.parametrize("name", golden_names(kind="homogeneity"))
def test_golden_homogeneity(name):
    """Caller of the hot path: P1/P2/P3 equal the reference's values; P3 is one batched launch."""
    from statdepth_amd.homogeneity import FunctionalHomogeneity
    fx = load_golden(name)
    F, G = frame_df(fx["input"]["F"]), frame_df(fx["input"]["G"])
    Fc, Gc = F.copy(), G.copy()
    h = FunctionalHomogeneity([F], [G], method=fx["call"]["method"], relax=True, quiet=True).homogeneity()
    got = np.asarray(h, dtype=float).ravel()
    assert_depths_close(got, np.array(fx["value"], dtype=float), TOL)
    assert F.equals(Fc) and G.equals(Gc)           # inputs are not mutated (the reference does mutate F)


def test_subset_counts_vs_oracle(eng, oracle):
    """sd_mbd_subset_counts: depth of a target inside an explicit block of curves == oracle on the sub-matrix."""
    rng = np.random.default_rng(41)
    X = np.round(rng.normal(size=(23, 90)).cumsum(axis=0), 1)
    X[4, 17] = np.nan
    blocks, tgs = [], []
    for k in range(40):
        size = int(rng.integers(3, 70))
        mem = rng.choice(90, size=size, replace=False)
        blocks.append(mem)
        tgs.append(int(mem[rng.integers(0, size)]))
    width = max(len(b) for b in blocks)
    M = np.full((len(blocks), width), -1, dtype=np.int32)
    for i, b in enumerate(blocks):
        M[i, :len(b)] = b
    for J in (2, 3):
        got = eng.mbd_subset_counts(X, M, np.array(tgs), J=J)
        for i, (b, tg) in enumerate(zip(blocks, tgs)):
            sub = X[:, b]
            want = oracle.mbd_counts(sub, [list(b).index(tg)], J)[0]
            assert (got[i] == want).all()


def test_external_counts_vs_oracle(eng, oracle):
    """sd_mbd_external_counts == oracle depth counts of g inside F u {g}, for every g at once."""
    rng = np.random.default_rng(12)
    F = np.round(rng.normal(size=(19, 300)).cumsum(axis=0), 1)
    G = np.round(rng.normal(size=(19, 70)).cumsum(axis=0) + 0.3, 1)
    G[3, 5] = np.nan
    F[7, 11] = np.nan
    for J in (2, 3):
        got = eng.mbd_external_counts(F, G, J=J)
        for q in (0, 5, 33, 69):
            Fg = np.concatenate([F, G[:, q:q + 1]], axis=1)
            want = oracle.mbd_counts(Fg, [F.shape[1]], J)[0]
            assert (got[q] == want).all()


@pytest.mark.parametrize("n,m", [(40, 20), (700, 33), (5000, 1500), (10000, 2500), (16384, 9000), (20000, 40)])
def test_external_counts_rows_and_sizes(eng, oracle, monkeypatch, n, m):
    """External targets through the per-row bucket structure (n <= 16384, m >= 16) and the pairwise kernel (otherwise):
    rows of every kind, targets inside / outside the set's range, tied with set values, NaN; several target groups."""
    rng = np.random.default_rng(n + m)
    T = 12
    F = rng.normal(size=(T, n)).cumsum(axis=0)
    G = rng.normal(size=(T, m)).cumsum(axis=0) * 1.5
    F[1] = np.round(F[1], 0); G[1] = np.round(G[1], 0)          # ties with set values, crowded buckets
    F[2, ::7] = np.nan; G[2, ::5] = np.nan
    F[3, 1] = np.inf; F[3, 2] = -np.inf; G[3, 0] = np.inf; G[3, 1] = -np.inf
    F[4, :] = 0.5; G[4, :3] = [0.5, 0.25, 0.75]                  # all set values equal: target tied / below / above
    F[5, :] = np.nan
    G[6, :] = np.nan
    G[7, :4] = [F[7].min() - 1.0, F[7].max() + 1.0, F[7].min(), F[7].max()]
    F[8, 1:] = np.nan                                            # a single value in the set
    F[9, :2] *= 1e7                                              # outlying set curves: robust range, tails clamp
    G[9, :2] = [F[9].max() * 2, F[9].min()]
    F[10, 3] = 1e300; G[10, 3] = -1e300
    tg = np.unique(np.concatenate([np.arange(min(m, 6)), rng.integers(0, m, size=6), [m - 1]]))
    for J in (2, 3):
        got = eng.mbd_external_counts(F, G, J=J)
        for q in tg:
            Fg = np.concatenate([F, G[:, q:q + 1]], axis=1)
            assert (got[q] == oracle.mbd_counts(Fg, [n], J)[0]).all(), (q, J)


# ---------------------------------------------------------------- randomised, against the oracle
def _cases():
    rng = np.random.default_rng(123)
    out = []
    for (T, n, kind) in [(1, 2, "normal"), (3, 5, "ints"), (17, 33, "normal"), (64, 64, "ints"),
                         (65, 257, "walk"), (100, 50, "normal"), (37, 1023, "ints"), (9, 1025, "walk"),
                         (130, 2049, "normal"), (5, 4097, "ints"), (3, 8200, "walk"), (2, 16384, "ints"),
                         (3, 16385, "ints"), (2, 33000, "walk"), (4, 40000, "normal")]:
        if kind == "normal":
            X = rng.normal(size=(T, n))
        elif kind == "ints":
            X = rng.integers(0, 7, size=(T, n)).astype(float)
        else:
            X = np.round(rng.normal(size=(T, n)).cumsum(axis=0), 1)
        out.append(X)
    return out


@pytest.mark.parametrize("X", _cases(), ids=lambda X: f"{X.shape[0]}x{X.shape[1]}")
@pytest.mark.parametrize("algo", ["pairwise", "rank"])
def test_mbd_counts_vs_oracle(eng, oracle, X, algo):
    T, n = X.shape
    for J in (2, 3):
        if n - 1 < J or (n > 20000 and J == 3):
            continue
        want = oracle.mbd_counts(X, None, J)
        assert (eng.mbd_counts(X, None, J, algo=algo) == want).all()
        # curve-major (F-contiguous) input: transposed on the device
        assert (eng.mbd_counts(np.asfortranarray(X), None, J, algo=algo) == want).all()
    tg = np.unique(np.random.default_rng(n).integers(0, n, size=min(n, 70)))[::-1].copy()
    assert (eng.mbd_counts(X, tg, 2, algo=algo) == oracle.mbd_counts(X, tg, 2)).all()


@pytest.mark.parametrize("algo", ["pairwise", "rank"])
def test_mbd_nan_inf_ties(eng, oracle, algo):
    rng = np.random.default_rng(77)
    X = rng.integers(-2, 3, size=(23, 301)).astype(float)
    X[rng.random(X.shape) < 0.05] = np.nan
    X[rng.random(X.shape) < 0.02] = np.inf
    X[rng.random(X.shape) < 0.02] = -np.inf
    X[rng.random(X.shape) < 0.02] = -0.0
    X[5, :] = 1.0            # a fully tied timepoint
    X[6, :] = np.nan         # an all-NaN timepoint
    X[:, 17] = X[:, 3]       # duplicated curve
    for J in (2, 3):
        assert (eng.mbd_counts(X, None, J, algo=algo) == oracle.mbd_counts(X, None, J)).all()


def test_mbd_counts_range_vs_oracle(eng, oracle):
    """Contiguous target block (the sharded path's form), all three kernels."""
    rng = np.random.default_rng(31)
    for (T, n, lo, m) in [(7, 500, 100, 250), (3, 20000, 9000, 10000), (2, 40000, 16000, 20000), (2, 40000, 0, 5)]:
        X = np.round(rng.normal(size=(T, n)).cumsum(axis=0), 1)
        want = oracle.mbd_counts(X, np.arange(lo, lo + m), 2)
        for algo in ("pairwise", "rank", "auto"):
            if algo == "pairwise" and n * m > 3e8:
                continue
            assert (eng.mbd_counts_range(X, lo, m, 2, algo=algo) == want).all()


def test_mbd_high_J(eng, oracle, xcheck):
    rng = np.random.default_rng(5)
    X = rng.integers(0, 9, size=(12, 40)).astype(float)
    X[2, 3] = np.nan
    for J in (4, 5, 8):
        for algo in ("pairwise", "rank"):
            assert (eng.mbd_counts(X, None, J, algo=algo) == oracle.mbd_counts(X, None, J)).all()
    # J >= 4 on the rank path: bucket kernel in image mode + fold, against the sort-based predecessor and the oracle
    Y = _bucket_rows(np.random.default_rng(6), 16, 3000)
    tg = np.arange(0, 3000, 111)
    for J in (4, 5):
        got = eng.mbd_counts(Y, None, J, algo="rank")
        with xcheck(SD_RANK_IMPL="3"):
            assert (got == eng.mbd_counts(Y, None, J, algo="rank")).all()
        assert (got[tg] == oracle.mbd_counts(Y, tg, J)).all()


def test_rank_implementations_cross_check(eng, oracle, xcheck):
    """The four rank implementations (bucket kernel, packed keys + deferred rows, search for every row, first generation)
    are independent pieces of code: they must agree with each other and with the oracle on ties, near-ties
    (values equal above the index field of the packed key) and specials."""
    rng = np.random.default_rng(99)
    base = np.round(rng.normal(size=(9, 3000)).cumsum(axis=0), 1)
    near = 1.0 + rng.integers(0, 50, size=(9, 3000)) * 2.0 ** -50           # distinct values, identical high bits
    big = 1.7e9 + rng.normal(size=(9, 3000)) * 1e-3                         # timestamps-like magnitudes
    tiny = rng.integers(-3, 4, size=(9, 3000)) * 5e-324                      # denormals around zero
    cont = rng.normal(size=(9, 3000))
    cont[4, :7] = [np.inf, -np.inf, np.nan, 0.0, -0.0, 1.79e308, -1.79e308]
    for X in (base, near, big, tiny, cont):
        want = oracle.mbd_counts(X, None, 2)
        assert (eng.mbd_counts(X, None, 2, algo="rank") == want).all(), "product library"
        for impl in ("4", "3", "2", "1"):
            with xcheck(SD_RANK_IMPL=impl):
                assert (eng.mbd_counts(X, None, 2, algo="rank") == want).all(), impl


def _bucket_rows(rng, T, n):
    """Rows that exercise every branch of the bucket kernel: plain, tied, NaN / +-inf, constant, crowded."""
    X = rng.normal(size=(T, n)).cumsum(axis=0)
    X[1] = np.round(X[1], 1)                         # ties inside buckets
    X[2] = np.round(X[2], 0)                         # heavy ties: crowded buckets -> sorted behind the loop
    X[3, ::9] = np.nan                               # NaNs stay out of the histogram
    X[4, 5 % n] = np.inf                             # an infinity sets the row aside
    X[4, 6 % n] = -np.inf
    X[5, :] = 1.25                                   # all values equal
    X[6, :] = np.nan                                 # nothing to rank
    X[7, 1:] = np.nan                                # a single value
    X[8] = np.exp(X[8])                              # skewed
    X[9] = rng.standard_cauchy(n)                    # heavy tails: nearly everything in a few buckets
    X[10, 0], X[10, n - 1] = 1.7e308, -1.7e308       # the range overflows
    X[11] = X[11] * 1e-310                           # denormals
    X[12, : n // 2] = X[12, n // 2: 2 * (n // 2)]    # pairs of equal values
    X[13, : min(3, n)] = [0.0, -0.0, 5e-324][: min(3, n)]
    return X


@pytest.mark.parametrize("n", [2, 3, 64, 65, 1000, 1025, 3000, 8192, 8193, 10000, 12345, 16384])
def test_bucket_kernel_rows(eng, oracle, xcheck, n):
    """mbd_rank_bucket.hip (the default for n <= 16384) against the packed-sort path of the cross-check build on every
    curve and against the oracle on a sample of targets."""
    rng = np.random.default_rng(n)
    X = _bucket_rows(rng, 16, n)
    for J in (2, 3):
        if n - 1 < J:
            continue
        got = eng.mbd_counts(X, None, J, algo="rank")
        with xcheck(SD_RANK_IMPL="3"):
            assert (got == eng.mbd_counts(X, None, J, algo="rank")).all()
        tg = np.unique(np.linspace(0, n - 1, 48).astype(np.int64))
        assert (got[tg] == oracle.mbd_counts(X, tg, J)).all()


def test_bucket_kernel_many_rows_per_workgroup(eng, oracle):
    """More rows than workgroups (several rows per workgroup, register accumulators across rows) with rows set aside
    in between."""
    rng = np.random.default_rng(8)
    T, n = 1500, 700
    X = rng.normal(size=(T, n)).cumsum(axis=0)
    X[::7] = np.round(X[::7], 0)
    X[5::11, 3] = np.inf
    X[2::13, ::4] = np.nan
    tg = np.arange(0, n, 9)
    assert (eng.mbd_counts(X, tg, 2, algo="rank") == oracle.mbd_counts(X, tg, 2)).all()
    assert (eng.mbd_counts(X, None, 2, algo="rank")[tg] == oracle.mbd_counts(X, tg, 2)).all()


@pytest.mark.parametrize("n", [700, 3000, 10000, 16384])
def test_bucket_kernel_tie_rows(eng, oracle, n):
    """Rows with big buckets: every bucket holds one value (closed form, no member pass), a crowded bucket with two
    values a hair apart (set aside for the sort), a moderately full mixed bucket (back to the member passes), with
    and without NaNs; in between ordinary rows."""
    rng = np.random.default_rng(n)
    T = 40
    X = rng.normal(size=(T, n)).cumsum(axis=0)
    X[0::8] = rng.integers(0, 6, size=X[0::8].shape).astype(float)            # pure, far above CAP keys per bucket
    X[1::8] = np.round(X[1::8], 1)                                            # pure, moderate buckets
    r2 = rng.integers(0, 4, size=X[2::8].shape).astype(float)
    r2[:, : n // 3] = np.where(rng.random((r2.shape[0], n // 3)) < 0.5, 1.0, 1.0 + 1e-13)
    X[2::8] = r2                                                              # crowded and mixed: the sort
    r3 = rng.uniform(0, 100, size=X[3::8].shape)
    r3[:, :40] = np.where(rng.random((r3.shape[0], 40)) < 0.5, 50.0, 50.0 + 1e-12)
    X[3::8] = r3                                                              # 40 keys, two values, one bucket
    X[4::8] = X[0::8]
    X[4::8][rng.random(X[4::8].shape) < 0.1] = np.nan                         # pure with NaNs
    X[5, :] = 3.0
    X[5, ::3] = np.nan
    X[6, :] = np.nan                                                          # no value at all: skipped
    tg = np.unique(rng.integers(0, n, size=60))
    for J in (2, 3, 4):
        got = eng.mbd_counts(X, None, J, algo="rank")[tg]
        assert (got == oracle.mbd_counts(X, tg, J)).all(), J


@pytest.mark.parametrize("n", [1500, 3000, 10000, 16384])
def test_bucket_kernel_outlier_rows(eng, oracle, n):
    """Far-out values stretch a row's range: the kernel takes the bracket of the waves' innermost extremes as the range
    and clamps the tails into the end buckets (monotone for any range).  A few outlying curves, infinities, an
    overflowing range, outliers in every wave, outliers tied with each other."""
    rng = np.random.default_rng(n + 5)
    T = 24
    X = rng.normal(size=(T, n)).cumsum(axis=0)
    X[:, :3] *= 1e6                                               # three outlying curves (one wave)
    X[2, 7] = np.inf
    X[2, 8] = -np.inf
    X[3, 9] = 1.7e308
    X[3, 10] = -1.7e308                                           # hi - lo overflows
    X[4, rng.choice(n, size=n // 50, replace=False)] *= 1e5       # 2 % outliers: every wave has some
    X[5, 20:60] = 1e9                                             # tied outliers
    X[6, 20:60] = -np.inf
    X[7, ::2] = np.nan
    X[8, 100:] = 0.25                                             # the bulk on one value, a few keys elsewhere
    tg = np.unique(np.concatenate([np.arange(12), np.arange(20, 24), rng.integers(0, n, size=40)]))
    for J in (2, 3, 4):
        got = eng.mbd_counts(X, None, J, algo="rank")[tg]
        assert (got == oracle.mbd_counts(X, tg, J)).all(), J


@pytest.mark.parametrize("T", [2100, 4100, 8200])
def test_bucket_kernel_accumulator_widths(eng, oracle, T):
    """n = 16000: 9 rows per workgroup -> 32-bit register accumulators and u32 partial totals; 17 rows -> 64-bit
    accumulators, u32 partials; 33 rows -> u64 throughout.  The totals near the middle ranks approach
    rows * C(n-1, 2) / 2, so a width chosen too small would wrap."""
    rng = np.random.default_rng(T)
    n = 16000
    X = rng.normal(size=(T, n))
    order = np.argsort(X[0])
    tg = np.concatenate([order[n // 2 - 2:n // 2 + 2], order[:2], order[-2:]])
    X[:, tg[:4]] *= 1e-3                                  # four curves that stay near the middle: the largest totals
    want = oracle.mbd_counts(X, tg, 2)
    got = eng.mbd_counts(X, None, 2, algo="rank")[tg]
    assert (got == want).all()
    assert int(want.max()) > T * (n - 1) * (n - 2) // 5


def test_above_below_vs_oracle(eng, oracle):
    rng = np.random.default_rng(9)
    X = rng.integers(0, 5, size=(21, 130)).astype(float)
    assert (eng.above_below(X) == oracle.above_below(X)).all()


@pytest.mark.parametrize("shape", [(5, 6), (32, 70), (33, 600), (64, 40), (65, 130), (128, 300), (129, 700), (200, 77),
                                   (1024, 300), (1030, 45)])
def test_strict_vs_oracle(eng, oracle, shape):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    T, n = shape
    # smooth, well separated curves so that a good share of pairs is contained at every t
    base = np.sort(rng.normal(size=n))[None, :] * 3.0 + rng.normal(size=(T, n)) * 0.3
    X = np.round(base, 1)
    X[:, n // 2] = X[:, 0]
    want = oracle.bd_strict_counts(X)
    assert want.sum() > 0
    assert (eng.bd_strict_counts(X)[:, 0] == want).all()
    assert (eng.bd_strict_counts(np.asfortranarray(X))[:, 0] == want).all()
    Xn = X.copy()
    Xn[rng.random(X.shape) < 0.03] = np.nan
    Xn[:, 1] = X[:, 1]
    assert (eng.bd_strict_counts(Xn)[:, 0] == oracle.bd_strict_counts(Xn)).all()


@pytest.mark.parametrize("shape", [(5, 6), (31, 70), (32, 600), (33, 257), (64, 40), (100, 1300), (999, 300), (1000, 700),
                                   (1024, 130), (1025, 45), (2100, 150)])
def test_strict_complement_matching_vs_oracle(eng, oracle, xcheck, shape):
    """Continuous data: every curve is strictly above or below the target at every timepoint, so the J = 2 count comes
    from grouping complementary masks (strict_match_*_kernel) instead of the pair walk.  Banded curves (many equal
    masks: whole groups above / below), crossing curves (unique masks), one exact mirror pair, and a duplicated curve
    (its two copies are the only targets that fall back to the pair kernel).  Beyond 1 024 timepoints the fallback is
    the first-generation pair kernel, gated per target."""
    rng = np.random.default_rng(shape[0] * 7919 + shape[1])
    T, n = shape
    X = np.sort(rng.normal(size=n))[None, :] * 3.0 + rng.normal(size=(T, n)) * 0.3
    X[:, 3] = 2.0 * X[:, 2].mean() - X[:, 2]          # mirror image of curve 2 about a constant
    if n > 40:
        X[:, n // 2] = X[:, 7]                        # exact duplicate: ties at every timepoint for these two targets
    want = oracle.bd_strict_counts(X)
    assert want.sum() > 0
    got = eng.bd_strict_counts(X)[:, 0]
    assert (got == want).all()
    assert (eng.bd_strict_counts(np.asfortranarray(X))[:, 0] == want).all()
    tg = rng.permutation(n)[: max(2, n // 3)]
    assert (eng.bd_strict_counts(X, tg)[:, 0] == want[tg]).all()
    with xcheck(SD_STRICT_NOMATCH=1):
        assert (eng.bd_strict_counts(X)[:, 0] == want).all()
    with xcheck(SD_STRICT_GLOBAL_TABLE=1):           # the table of n > 16 384, in global memory
        assert (eng.bd_strict_counts(X)[:, 0] == want).all()
    with xcheck(SD_STRICT_FP64_MASKS=1):             # masks from the values instead of the rank image (n > 16 384)
        assert (eng.bd_strict_counts(X)[:, 0] == want).all()


def test_strict_complement_matching_random_walks_vs_pair_kernel(eng, xcheck):
    """2 000 random walks x 1 000 timepoints (the size the strict timings are quoted on): matching path against the
    pair kernel of the cross-check library, plus curves built to be exact complements of each other."""
    rng = np.random.default_rng(12)
    T, n = 1000, 2000
    X = rng.normal(size=(T, n)).cumsum(axis=0)
    X[:, 100:160] += 400.0                            # a group always above everything
    X[:, 200:250] -= 400.0                            # and one always below
    got = eng.bd_strict_counts(X)[:, 0]
    with xcheck(SD_STRICT_NOMATCH=1):
        want = eng.bd_strict_counts(X)[:, 0]
    assert (got == want).all()
    assert got.sum() > 0


@pytest.mark.parametrize("shape", [(40, 300), (1000, 600), (1500, 200)])
def test_strict_matching_with_common_values_and_sparse_ties(eng, oracle, xcheck, shape):
    """The cases between 'all clean' and 'all ties': every curve shares its value at some timepoints (a common start:
    those timepoints are dropped from the matching), a few curves touch other curves or hold NaN (only pairs with
    such a curve go through the pair kernel; beyond 1 024 timepoints their targets do as a whole)."""
    rng = np.random.default_rng(shape[0] + shape[1])
    T, n = shape
    X = np.sort(rng.normal(size=n))[None, :] * 3.0 + rng.normal(size=(T, n)) * 0.3
    X[:, 3] = 2.0 * X[:, 2].mean() - X[:, 2]
    X[0, :] = 0.0                                     # common start
    X[T // 2, :] = 1.25                               # and a common value mid-way
    Xa = X.copy()
    want = oracle.bd_strict_counts(Xa)
    assert want.sum() > 0
    assert (eng.bd_strict_counts(Xa)[:, 0] == want).all()
    # sparse ties: curve 5 touches curve 9 at three timepoints, curve 11 duplicates curve 12, curve 20 has NaN
    Xb = X.copy()
    Xb[[3, 7, T - 1], 5] = Xb[[3, 7, T - 1], 9]
    Xb[:, 11] = Xb[:, 12]
    Xb[T // 3, 20] = np.nan
    wantb = oracle.bd_strict_counts(Xb)
    gotb = eng.bd_strict_counts(Xb)[:, 0]
    assert (gotb == wantb).all()
    with xcheck(SD_STRICT_NOMATCH=1):
        assert (eng.bd_strict_counts(Xb)[:, 0] == wantb).all()
    with xcheck(SD_STRICT_GLOBAL_TABLE=1):
        assert (eng.bd_strict_counts(Xb)[:, 0] == wantb).all()
    # every row constant: every pair contains every target
    Xc = np.tile(rng.normal(size=(T, 1)), (1, 9))
    assert (eng.bd_strict_counts(Xc)[:, 0] == 28).all()
    assert (oracle.bd_strict_counts(Xc) == 28).all()


def test_strict_matching_groups_with_many_members(eng, oracle, xcheck):
    """Several curves with the SAME crossing pattern relative to a target, on both sides: the group counters
    (count(side 0) * count(side 1) per canonical mask) rather than groups of one."""
    rng = np.random.default_rng(77)
    T, n = 96, 60
    tgt = rng.normal(size=T)
    sign = np.where(rng.random(T) < 0.5, 1.0, -1.0)           # one crossing pattern
    X = np.empty((T, n))
    X[:, 0] = tgt
    for i in range(1, 8):
        X[:, i] = tgt + sign * (0.1 + 0.05 * i)               # seven curves following the pattern
    for i in range(8, 13):
        X[:, i] = tgt - sign * (0.2 + 0.03 * i)               # five following its complement
    sign2 = np.where(rng.random(T) < 0.5, 1.0, -1.0)
    for i in range(13, 16):
        X[:, i] = tgt + sign2 * (0.15 + 0.01 * i)
    for i in range(16, 20):
        X[:, i] = tgt - sign2 * (0.3 + 0.01 * i)
    X[:, 20:] = tgt[:, None] + rng.normal(size=(T, n - 20)) * 2.0 + np.linspace(-30, 30, n - 20)[None, :]
    want = oracle.bd_strict_counts(X)
    assert want[0] >= 7 * 5 + 3 * 4
    assert (eng.bd_strict_counts(X)[:, 0] == want).all()
    with xcheck(SD_STRICT_GLOBAL_TABLE=1):
        assert (eng.bd_strict_counts(X)[:, 0] == want).all()


def test_strict_masks_from_ranks_with_nan_inf_and_ties(eng, oracle, xcheck):
    """The mask kernel works on the bucket kernel's rank image (B = curves strictly below): NaN entries, +-inf, ties,
    an all-NaN timepoint and an all-equal timepoint, against the oracle and against the fp64 mask kernel."""
    rng = np.random.default_rng(5)
    T, n = 70, 333
    X = np.round(rng.normal(size=(T, n)).cumsum(axis=0), 1)
    X[rng.random(X.shape) < 0.02] = np.nan
    X[3, 10:20] = np.inf
    X[5, 30:35] = -np.inf
    X[7, 40] = np.inf
    X[9, :] = np.nan
    X[11, :] = 2.0
    X[:, 50] = np.linspace(-3, 3, T)                  # NaN-free curves so that some targets count
    X[:, 51] = X[:, 50] + 100.0
    X[:, 52] = X[:, 50] - 100.0
    X[9, 50:53] = [0.0, 100.0, -100.0]
    want = oracle.bd_strict_counts(X)
    got = eng.bd_strict_counts(X)[:, 0]
    assert (got == want).all()
    with xcheck(SD_STRICT_FP64_MASKS=1):
        assert (eng.bd_strict_counts(X)[:, 0] == want).all()
    Xi = X.copy()
    Xi[9, :] = 1.0                                    # without the all-NaN timepoint more targets count
    Xi[np.isnan(Xi)] = 0.5
    wanti = oracle.bd_strict_counts(Xi)
    assert wanti.sum() > 0
    assert (eng.bd_strict_counts(Xi)[:, 0] == wanti).all()


def test_strict_J3_J4_vs_literal_enumeration(eng, oracle):
    rng = np.random.default_rng(4)
    X = np.round(np.sort(rng.normal(size=12))[None, :] * 2 + rng.normal(size=(6, 12)) * 0.4, 1)
    for J in (3, 4):
        assert (eng.bd_strict_counts(X, None, J) == oracle.band_enum(X, None, J, relax=False)).all()


def test_l1_vs_oracle(eng, oracle):
    rng = np.random.default_rng(2)
    for (n, d) in [(5, 2), (300, 3), (257, 8), (64, 11)]:
        P = rng.normal(size=(n, d))
        got, want = eng.l1_depth(P), oracle.l1_depth(P)
        assert_depths_close(got, want, TOL)
    P = rng.normal(size=(40, 3))
    P[7] = P[3]
    got = eng.l1_depth(P)
    assert np.isnan(got[7]) and np.isnan(got[3]) and np.isnan(got).sum() == 2


def test_simplex_vs_oracle(eng, oracle):
    rng = np.random.default_rng(6)
    for (n, d) in [(25, 2), (14, 3), (12, 4), (9, 1)]:
        P = rng.normal(size=(n, d))
        assert (eng.pointcloud_simplex_counts(P) == oracle.pointcloud_simplex_counts(P)).all()
    P = rng.integers(0, 3, size=(14, 2)).astype(float)       # lattice: degenerate + boundary cases
    assert (eng.pointcloud_simplex_counts(P) == oracle.pointcloud_simplex_counts(P)).all()
    C = rng.normal(size=(8, 5, 2))
    for relax in (True, False):
        assert (eng.multi_simplex_counts(C, None, relax) == oracle.multi_simplex_counts(C, None, relax)).all()


def test_sampled_simplex_estimators(eng, oracle):
    """Configs 4/5 use the build's seeded subset sampler: kernel == CPU restatement draw for draw, and the
    estimate converges to the exhaustive count."""
    rng = np.random.default_rng(8)
    P = rng.normal(size=(40, 3))
    for seed in (0, 12345):
        got = eng.pointcloud_simplex_counts(P, samples=500, seed=seed)
        assert (got == oracle.simplex_sampled(P, samples=500, seed=seed)).all()
    C = rng.normal(size=(12, 6, 2)).cumsum(axis=1)
    for relax in (True, False):
        got = eng.multi_simplex_counts(C, [0, 5, 11], relax=relax, samples=300, seed=7)
        assert (got == oracle.simplex_sampled(C, [0, 5, 11], relax=relax, samples=300, seed=7)).all()
    # statistical agreement with the exhaustive depth: P(contain) estimated from 20000 draws
    Q = rng.normal(size=(25, 2))
    import math
    exact = oracle.pointcloud_simplex_counts(Q) / math.comb(24, 3)
    est = eng.pointcloud_simplex_counts(Q, samples=20000, seed=3) / 20000.0
    assert np.max(np.abs(est - exact)) < 0.02


# ---------------------------------------------------------------- BASELINE.json sizes: size-independent properties
def test_config2_scale_properties(eng, oracle):
    """10 000 curves x 1 000 timepoints (BASELINE.json configs[1]): the two HIP formulations agree on
    a target subset, the oracle agrees on a smaller one, and the counts obey their invariants."""
    rng = np.random.default_rng(1234)
    T, n = 1000, 10000
    X = rng.normal(size=(T, n)).cumsum(axis=0)
    full = eng.mbd_counts(X, None, 2, algo="rank")[:, 0]
    tg = np.arange(0, n, 97)
    assert (eng.mbd_counts(X, tg, 2, algo="pairwise")[:, 0] == full[tg]).all()
    tg2 = np.array([0, 1, 4999, 9999])
    assert (oracle.mbd_counts(X, tg2, 2)[:, 0] == full[tg2]).all()
    # invariants: 0 <= count <= T*C(n-1,2); sum over curves of (A - B) is 0 at every t, which for J=2 gives
    # sum_i count_i = T*n*C(n-1,2) - 2*sum_{t,i} C(A,2) -- checked through the permutation symmetry instead:
    perm = rng.permutation(n)
    again = eng.mbd_counts(np.ascontiguousarray(X[:, perm]), None, 2, algo="rank")[:, 0]
    assert (again == full[perm]).all()
    cmax = T * (n - 1) * (n - 2) // 2
    assert full.min() >= 0 and full.max() <= cmax
    # without ties, ranks at each t are a permutation: sum_i [C(A,2)+C(B,2)] is the same at every t
    expect = T * (n * cmax // T - 2 * sum(k * (k - 1) // 2 for k in range(n)))
    assert int(full.sum()) == expect


# ---------------------------------------------------------------- edge cases of the boundary
def test_edge_cases_api(oracle):
    from statdepth_amd import FunctionalDepth, PointcloudDepth
    rng = np.random.default_rng(3)
    # smallest legal problem: 2 timepoints... J must be < number of rows, so T >= 3 for J = 2
    df = pd.DataFrame(rng.normal(size=(3, 3)), columns=["a", "b", "c"], index=[10, 20, 30])
    for relax in (True, False):
        got = FunctionalDepth([df], relax=relax)
        assert_depths_close(got.to_numpy(), oracle.univariate_depths(df.to_numpy(), None, 2, relax), TOL)
    # empty to_compute -> empty Series
    e = FunctionalDepth([df], to_compute=[], relax=True)
    assert len(e) == 0
    # duplicated / reordered targets keep the caller's order
    d = FunctionalDepth([df], to_compute=["c", "a", "c"], relax=True)
    assert list(d.index) == ["c", "a", "c"] and d.iloc[0] == d.iloc[2]
    # integer dtype frames and constant data (everything tied): every band contains everything
    di = pd.DataFrame(np.ones((4, 6), dtype=np.int64))
    assert_depths_close(FunctionalDepth([di], relax=True).to_numpy(), oracle.univariate_depths(np.ones((4, 6)), None, 2, True), TOL)
    assert_depths_close(FunctionalDepth([di], relax=False).to_numpy(), oracle.univariate_depths(np.ones((4, 6)), None, 2, False), TOL)
    # J larger than the number of other curves: those terms vanish (C(n-1, J) = 0 subsets)
    dj = pd.DataFrame(rng.normal(size=(6, 3)))
    assert_depths_close(FunctionalDepth([dj], J=3, relax=True).to_numpy(),
                        oracle.univariate_depths(dj.to_numpy(), None, 3, True), TOL)
    # all-NaN curve, and huge / tiny magnitudes
    dn = pd.DataFrame(rng.normal(size=(5, 7)))
    dn.iloc[:, 2] = np.nan
    dn.iloc[1, 4] = 1e308
    dn.iloc[2, 5] = -1e-320
    for relax in (True, False):
        assert_depths_close(FunctionalDepth([dn], relax=relax).to_numpy(),
                            oracle.univariate_depths(dn.to_numpy(), None, 2, relax), TOL)
    # point cloud with exactly d + 2 points and labelled index
    pc = pd.DataFrame(rng.normal(size=(4, 2)), index=list("wxyz"))
    got = PointcloudDepth(pc)
    assert list(got.index) == list("wxyz")
    assert_depths_close(got.to_numpy(), oracle.pointcloud_depths(pc.to_numpy()), TOL)
    assert len(PointcloudDepth(pc, to_compute=[], containment="l1")) == 0


def test_overflow_is_reported_not_wrapped():
    """T*C(n-1,J) >= 2^63 must raise instead of returning wrapped integers."""
    from statdepth_amd import engine
    from statdepth_amd._native import StatdepthHipError, SD_ERR_OVERFLOW
    import torch
    X = torch.zeros((5, 200000), dtype=torch.float64, device="cuda")
    with pytest.raises(StatdepthHipError) as ei:
        engine.mbd_counts(X, [0, 1], J=4, algo="pairwise")
    assert ei.value.code == SD_ERR_OVERFLOW


def test_config3_scale_properties(eng, oracle):
    """100 000 curves x 256 timepoints (BASELINE.json configs[2]) on one GPU: the chunked rank kernels against the
    pairwise kernel on a target subset and the oracle on a few targets, plus the no-tie rank-sum invariant."""
    import torch
    T, n = 256, 100000
    g = torch.Generator(device="cuda").manual_seed(1235)
    X = torch.randn(T, n, dtype=torch.float64, device="cuda", generator=g).cumsum(0)
    full = eng.mbd_counts(X, None, 2, algo="rank")[:, 0]
    tg = np.arange(5, n, 997)
    assert (eng.mbd_counts(X, tg, 2, algo="pairwise")[:, 0] == full[tg]).all()
    Xh = X.cpu().numpy()
    tg2 = np.array([0, 16383, 16384, 50000, 99999])
    assert (oracle.mbd_counts(Xh, tg2, 2)[:, 0] == full[tg2]).all()
    expect = T * (n * ((n - 1) * (n - 2) // 2) - 2 * sum(k * (k - 1) // 2 for k in range(n)))
    assert int(full.astype(object).sum()) == expect
    # contiguous target block (what a rank of the sharded path asks for)
    blk = eng.mbd_counts_range(X, 37500, 12500, 2)[:, 0]
    assert (blk == full[37500:50000]).all()


def test_million_points_linf_shape(eng, oracle):
    """BASELINE.json configs[4] in its L-infinity form: 10^6 points in R^3 = 10^6 "curves" x 3 "timepoints"
    (SURVEY.md 8 P4): the large-n route with 182 value buckets per row, against the oracle and the pairwise kernel."""
    rng = np.random.default_rng(1237)
    X = np.ascontiguousarray(rng.normal(size=(1_000_000, 3)).T)
    tg = np.array([0, 1, 499_999, 777_777, 999_999, 123_456])
    want = oracle.mbd_counts(X, tg, 2)
    assert (eng.mbd_counts(X, tg, 2, algo="rank") == want).all()
    assert (eng.mbd_counts(X, tg, 2, algo="pairwise") == want).all()
    Xt = np.round(X, 2)                                    # ties: ~600 distinct values per coordinate
    assert (eng.mbd_counts(Xt, tg, 2, algo="rank") == oracle.mbd_counts(Xt, tg, 2)).all()


def test_sharded_paths_world_size_one(eng, oracle):
    """The multi-GPU paths on RCCL with a single rank: all-to-all / reduce-scatter (curves) and all-gather (points) run,
    the HIP engine is the compute hook, results equal the single-process calls."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from statdepth_amd.distributed import sharded_mbd_counts, sharded_pointcloud
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    try:
        rng = np.random.default_rng(12)
        X = np.round(rng.normal(size=(23, 500)).cumsum(axis=0), 1)
        Xd = torch.from_numpy(X).cuda()
        want = oracle.mbd_counts(X, None, 2)
        for mode in ("time", "targets"):
            assert (sharded_mbd_counts(Xd, J=2, mode=mode).cpu().numpy() == want).all()
        # world size 1 needs no exchange; force it so that RCCL's all-to-all / reduce-scatter run on the GPU
        assert (sharded_mbd_counts(Xd, J=2, mode="time", _force_exchange=True).cpu().numpy() == want).all()
        from statdepth_amd.distributed import sharded_bd_strict_counts
        Xs = X[:, :60]
        got = sharded_bd_strict_counts(torch.from_numpy(np.ascontiguousarray(Xs)).cuda(), J=2).cpu().numpy()
        assert (got[:, 0] == oracle.bd_strict_counts(Xs)).all()
        P = rng.normal(size=(40, 3))
        Pd = torch.from_numpy(P).cuda()
        got, n = sharded_pointcloud(Pd, "simplex")
        assert n == 40 and (np.asarray(got) == eng.pointcloud_simplex_counts(P)).all()
        got, _ = sharded_pointcloud(Pd, "simplex", samples=64, seed=5)
        assert (np.asarray(got) == oracle.simplex_sampled(P, np.arange(40), samples=64, seed=5)).all()
        got, _ = sharded_pointcloud(Pd, "l1")
        assert np.max(np.abs(np.asarray(got) - oracle.l1_depth(P))) <= TOL
    finally:
        dist.destroy_process_group()


def test_row_batching(eng, oracle, xcheck):
    """Rows are processed in batches when the pair image / sorted scratch would exceed 1 GiB; the cross-check build of
    the same launchers lets a switch force small batches."""
    rng = np.random.default_rng(77)
    X = np.round(rng.normal(size=(40, 700)).cumsum(axis=0), 1)
    X[11, 5] = np.nan
    with xcheck(SD_RANK_ROWS_PER_BATCH="7"):
        for J in (2, 3, 4):
            assert (eng.mbd_counts(X, None, J, algo="rank") == oracle.mbd_counts(X, None, J)).all()
        tg = np.array([3, 699, 0, 350])
        assert (eng.mbd_counts(X, tg, 2, algo="rank") == oracle.mbd_counts(X, tg, 2)).all()
    Xb = rng.normal(size=(5, 17000))
    with xcheck(SD_RANK_ROWS_PER_BATCH="2"):
        assert (eng.mbd_counts(Xb, None, 2, algo="rank") == oracle.mbd_counts(Xb, None, 2)).all()



def test_big_n_tie_rows(eng, oracle):
    """n > 16384, tie-heavy rows inside the value buckets: one value per fine bucket (closed form), a crowded fine
    bucket mixing two values a hair apart (search kernel), a moderately full mixed fine bucket (member passes)."""
    rng = np.random.default_rng(99)
    T, n = 6, 40000
    X = rng.normal(size=(T, n)).cumsum(axis=0)
    X[0] = np.round(X[0] * 10, 0)                                  # ~100 distinct values
    X[1] = rng.integers(0, 1000, size=n).astype(float)
    X[2, : n // 2] = np.where(rng.random(n // 2) < 0.5, 1.0, 1.0 + 1e-13)
    X[3, :40] = np.where(rng.random(40) < 0.5, 0.5, 0.5 + 1e-14)
    X[4] = np.round(X[4], 1)
    X[4, ::9] = np.nan
    tg = np.unique(rng.integers(0, n, size=50))
    for J in (2, 3):
        assert (eng.mbd_counts(X, None, J, algo="rank")[tg] == oracle.mbd_counts(X, tg, J)).all(), J


def test_big_n_routes(eng, oracle, xcheck):
    """n > 16384: value-bucket route (default), its overflow fallback (a row where most values tie cannot be cut
    into buckets of 8192) and the chunked route forced for every row: identical integers."""
    rng = np.random.default_rng(2024)
    T, n = 4, 30000
    X = rng.normal(size=(T, n)).cumsum(axis=0)
    X[1, rng.random(n) < 0.7] = 0.25            # 70 % of row 1 tied on one value: bucket overflow -> chunked fallback
    X[2] = np.round(X[2], 1)                    # tie-heavy row: buckets flagged for the search kernel
    X[3, ::7] = np.nan
    X[3, 5] = np.inf
    X[3, 6] = -np.inf
    want = oracle.mbd_counts(X, None, 2)
    assert (eng.mbd_counts(X, None, 2, algo="rank") == want).all()
    for switch in ({"SD_BIG_IMPL": "1"}, {"SD_BIG_SORT": "1"}, {"SD_BIG_PART1": "1"}):
        with xcheck(**switch):
            assert (eng.mbd_counts(X, None, 2, algo="rank") == want).all(), switch
    tg = np.array([0, 29999, 12345, 7])
    assert (eng.mbd_counts(X, tg, 3, algo="rank") == oracle.mbd_counts(X, tg, 3)).all()


# ---------------------------------------------------------------- simplex containment, d = 5..8 (config 4's kernels)
def _simplex_cloud(rng, n, d, kind):
    """Point sets for the register-resident simplex kernels: generic position, a hyperplane (every simplex rank
    deficient -> the "undecided" hand-off to the generic code), and a mix with repeated points."""
    P = rng.normal(size=(n, d))
    P[0] = P[1:].mean(axis=0)                                # a central point: counts that are not all zero
    if kind == "flat":
        P[:, d - 1] = 0.25                                   # all points in one hyperplane
    elif kind == "mixed":
        P[: n // 2, d - 1] = 0.0                             # half the cloud in a hyperplane
        P[1] = P[0]                                          # a repeated point
        P[n - 1] = 0.5 * (P[2] + P[3])                       # a point on a segment between two others
    return P


@pytest.mark.parametrize("d", [5, 6, 7, 8])
def test_simplex_high_d_exhaustive_vs_oracle(eng, oracle, xcheck, d):
    """simplex_kernel_fast<5..8> (BASELINE config 4's d = 8 instantiation among them): exhaustive enumeration at small
    n against the oracle and against the generic kernel, for point clouds and for multivariate curves."""
    rng = np.random.default_rng(100 + d)
    n = d + 4                                                # C(n-1, d+1) = C(d+3, 2) subsets per target
    for kind in ("generic", "flat", "mixed"):
        P = _simplex_cloud(rng, n, d, kind)
        want = oracle.pointcloud_simplex_counts(P)
        got = eng.pointcloud_simplex_counts(P)
        assert (got == want).all(), (d, kind)
        with xcheck(SD_SIMPLEX_GENERIC="1"):
            assert (eng.pointcloud_simplex_counts(P) == want).all(), (d, kind, "generic kernel")
        assert want[0] > 0, (d, kind)
    # curves that keep their relative position over time (strict containment happens) plus one wandering feature
    C = rng.normal(size=(d + 6, 1, d)) + 1e-3 * rng.normal(size=(d + 6, 4, d))
    C[0] = C[1:].mean(axis=0)
    C[2, :, d - 1] = C[3, :, d - 1]                          # two curves agree in one feature
    C[4, 3, :] += 5.0                                        # one curve leaves at the last timepoint
    for relax in (True, False):
        want = oracle.multi_simplex_counts(C, None, relax)
        assert want[0] > 0, (d, relax)
        assert (eng.multi_simplex_counts(C, None, relax) == want).all(), (d, relax)
    tg = np.array([d + 5, 0, 3])
    assert (eng.multi_simplex_counts(C, tg, True) == oracle.multi_simplex_counts(C, tg, True)).all()


@pytest.mark.parametrize("d", [5, 6, 7, 8])
def test_simplex_high_d_sampled_vs_oracle(eng, oracle, xcheck, d):
    """The seeded subset sampler at d = 5..8, kernel == CPU restatement draw for draw (fast and generic kernels)."""
    rng = np.random.default_rng(200 + d)
    P = rng.normal(size=(60, d)) * rng.uniform(0.5, 2.0, size=d)
    P[:5] *= 0.05                                            # deep points: non-zero counts
    want = oracle.simplex_sampled(P, samples=400, seed=d)
    assert want[:5].sum() > 0
    assert (eng.pointcloud_simplex_counts(P, samples=400, seed=d) == want).all()
    C = rng.normal(size=(40, 12, d)).cumsum(axis=1)
    C[:4] *= 0.05
    tg = np.array([0, 1, 2, 3, 17, 39])
    for relax in (True, False):
        want = oracle.simplex_sampled(C, tg, relax=relax, samples=200, seed=11)
        assert (eng.multi_simplex_counts(C, tg, relax=relax, samples=200, seed=11) == want).all(), (d, relax)
    want = oracle.simplex_sampled(C, tg, relax=True, samples=200, seed=11)
    with xcheck(SD_SIMPLEX_GENERIC="1"):
        assert (eng.multi_simplex_counts(C, tg, relax=True, samples=200, seed=11) == want).all()
    assert want[:4].sum() > 0


def test_config4_full_size(eng, oracle):
    """BASELINE.json configs[3]: 5 000 multivariate curves (d = 8) x 500 timepoints, seeded sampled simplicial depth with
    S = 4 096 subsets per target: the oracle on 8 targets, determinism, and independence
    of the way the targets are split over calls (what sharding over GPUs relies on)."""
    import torch
    n, T, d, S, seed = 5000, 500, 8, 4096, 1236
    g = torch.Generator(device="cuda").manual_seed(seed)
    P = torch.randn(n, T, d, dtype=torch.float64, device="cuda", generator=g).cumsum(1)
    P[:16] *= 0.02                                           # a few central curves: counts that are not all zero
    full = eng.multi_simplex_counts(P, None, relax=True, samples=S, seed=seed)
    assert full.shape == (n,) and full.min() >= 0 and full.max() <= S * T
    assert full[:16].sum() > 0
    tg = np.array([0, 3, 15, 16, 777, 2500, 4998, 4999])
    Ph = P.cpu().numpy()
    want = oracle.simplex_sampled(Ph, tg, relax=True, samples=S, seed=seed)
    assert (full[tg] == want).all()
    # split-independence: any block of targets computed on its own gives the rows of the full call
    for lo, hi in ((0, 625), (4375, 5000)):
        blk = eng.multi_simplex_counts(P, np.arange(lo, hi), relax=True, samples=S, seed=seed)
        assert (blk == full[lo:hi]).all()
    # strict form (c // T) on the central curves
    tgs = np.arange(0, 24)
    strict = eng.multi_simplex_counts(P, tgs, relax=False, samples=S, seed=seed)
    assert (strict[:8] == oracle.simplex_sampled(Ph, tgs[:8], relax=False, samples=S, seed=seed)).all()
    assert (strict <= full[tgs] // T).all()
    # the estimator through the public API shape: depth in [0, 1]
    depth = full.astype(np.float64) / T / S
    assert depth.max() <= 1.0


def test_config5_full_size_simplex_and_l1(eng, oracle):
    """BASELINE.json configs[4]: 10^6 points in R^3 -- sampled simplicial depth (R = 4 096 tetrahedra per point) and L1
    depth at full size, the oracle on a sample of targets (the L-infinity form is test_million_points_linf_shape)."""
    rng = np.random.default_rng(1237)
    n, R = 1_000_000, 4096
    P = rng.normal(size=(n, 3))
    tg = np.concatenate([np.arange(8), rng.integers(0, n, size=180), [n - 1]])
    got = eng.pointcloud_simplex_counts(P, samples=R, seed=1237)
    assert got.shape == (n,) and got.max() <= R
    want = oracle.simplex_sampled(P, tg, samples=R, seed=1237)
    assert (got[tg] == want).all()
    # central points are deep, far points are not: the estimate orders them like the distance from the origin
    r = np.linalg.norm(P, axis=1)
    assert got[r < 0.3].mean() > 4 * got[r > 2.5].mean()
    again = eng.pointcloud_simplex_counts(P, tg, samples=R, seed=1237)
    assert (again == want).all()
    l1 = eng.l1_depth(P)
    tl = tg[:64]
    assert_depths_close(l1[tl], oracle.l1_depth(P, tl), TOL)
    assert l1.min() >= 0.0 and l1.max() <= 1.0 and l1[np.argmin(r)] > 0.9


# ---------------------------------------------------------------- external targets / explicit blocks of points
def test_pointcloud_external_and_subsets_vs_oracle(eng, oracle):
    rng = np.random.default_rng(17)
    for (n, d, m) in [(14, 2, 9), (11, 3, 5), (9, 5, 4), (30, 1, 7)]:
        F = rng.normal(size=(n, d))
        G = rng.normal(size=(m, d)) * 0.5
        G[0] = F[0]                                          # an external point equal to a sample point
        cnt = eng.pointcloud_simplex_external_counts(F, G)
        l1 = eng.l1_external_depth(F, G)
        for q in range(m):
            Fg = np.concatenate([F, G[q:q + 1]])
            assert cnt[q] == oracle.pointcloud_simplex_counts(Fg, [n])[0], (n, d, q)
            assert_depths_close(l1[q:q + 1], oracle.l1_depth(Fg, [n]), TOL)
        # explicit blocks: others first, target last, ragged sizes
        blocks = []
        for k in range(25):
            size = int(rng.integers(d + 2, n + 1))
            blocks.append(rng.choice(n, size=size, replace=False))
        width = max(len(b) for b in blocks)
        M = np.full((len(blocks), width), -1, dtype=np.int32)
        for i, b in enumerate(blocks):
            M[i, :len(b)] = b
        cnt = eng.pointcloud_simplex_subset_counts(F, M)
        l1 = eng.l1_subset_depth(F, M)
        for i, b in enumerate(blocks):
            sub = F[b]
            assert cnt[i] == oracle.pointcloud_simplex_counts(sub, [len(b) - 1])[0], (n, d, i)
            assert_depths_close(l1[i:i + 1], oracle.l1_depth(sub, [len(b) - 1]), TOL)


@pytest.mark.parametrize("containment", ["simplex", "l1"])
def test_pointcloud_ksampled_vs_restatement(oracle, containment):
    """PointcloudDepth(K=...) (_samplepointwisedepth, _pointcloud.py:68-123; the reference itself cannot run it on
    pandas >= 2): the batched launch against a CPU restatement of the same draw sequence -- per point, ss = n // K
    draws of `data.sample(n=ss)`, the point appended when missed, exact depth inside the sample, mean."""
    from statdepth_amd import PointcloudDepth
    rng = np.random.default_rng(23)
    n, d, K = 26, 2, 3
    df = pd.DataFrame(rng.normal(size=(n, d)), index=[f"p{i}" for i in range(n)], columns=["x", "y"])
    tc = ["p3", "p0", "p25", "p11"]
    np.random.seed(99)
    got = PointcloudDepth(df, to_compute=tc, K=K, containment=containment)
    assert list(got.index) == tc
    np.random.seed(99)
    ss = n // K
    want = []
    for lab in tc:
        vals = []
        for _ in range(ss):
            sdata = df.sample(n=ss, axis=0)
            if lab not in sdata.index:
                sdata = pd.concat([sdata, df.loc[[lab], :]])
            pos = list(sdata.index).index(lab)
            if containment == "simplex":
                vals.append(oracle.pointcloud_depths(sdata.to_numpy(), [pos])[0])
            else:
                vals.append(oracle.l1_depth(sdata.to_numpy(), [pos])[0])
        want.append(np.mean(vals))
    assert_depths_close(got.to_numpy(), np.array(want), TOL)
    # every point (to_compute=None), K = 1 falls through to the exact depth
    np.random.seed(5)
    allp = PointcloudDepth(df, K=2, containment=containment)
    assert list(allp.index) == list(df.index) and np.isfinite(allp.to_numpy()).all()
    exact = PointcloudDepth(df, K=1, containment=containment)
    assert_depths_close(exact.to_numpy(), PointcloudDepth(df, containment=containment).to_numpy(), TOL)


# ---------------------------------------------------------------- homogeneity entry points
@pytest.mark.parametrize("name", golden_names(kind="homogeneity_fn"))
def test_golden_homogeneity_functions(name):
    """P1_homogeneity / P2_homogeneity (homogeneity.py:214-306) equal the reference's values, including P2's second term
    taken over F WITH the 'G_deepest' column the reference's P1 leaves behind; the caller's frames are not touched."""
    from statdepth_amd.homogeneity.homogeneity import P1_homogeneity, P2_homogeneity
    fx = load_golden(name)
    F, G = frame_df(fx["input"]["F"]), frame_df(fx["input"]["G"])
    Fc, Gc = F.copy(), G.copy()
    fn = P1_homogeneity if fx["call"]["fn"] == "P1" else P2_homogeneity
    got = fn(F, G, relax=fx["call"]["relax"], quiet=True)
    assert_depths_close(np.array([got], dtype=float), np.array(fx["value"], dtype=float), TOL)
    assert F.equals(Fc) and G.equals(Gc)


@pytest.mark.parametrize("relax", [True, False])
def test_homogeneity_p3_overlapping_labels(oracle, relax):
    """P3 with the default integer labels (F and G share them).  The reference overwrites F's own column with g and
    drops it, shrinking F as its loop runs (homogeneity.py:125-128; its value is kept in the fixture for the record).
    The build evaluates every g inside the intact F u {g} (n_F + 1 curves) on both of its paths -- the batched one
    (relax=True) and the call-by-call one (relax=False) -- pinned here against the oracle."""
    from statdepth_amd.homogeneity import FunctionalHomogeneity
    fx = load_golden("h_p3_overlap")
    F, G = frame_df(fx["input"]["F"]), frame_df(fx["input"]["G"])
    assert set(F.columns) & set(G.columns)
    Fc = F.copy()
    got = FunctionalHomogeneity([F], [G], method="p3", relax=relax, quiet=True).homogeneity()
    Fx, Gx = F.to_numpy(), G.to_numpy()
    nF = Fx.shape[1]
    in_F = [oracle.univariate_depths(np.concatenate([Fx, Gx[:, c:c + 1]], axis=1), [nF], 2, relax)[0]
            for c in range(Gx.shape[1])]
    g_depths = oracle.univariate_depths(Gx, None, 2, relax)
    want = max(in_F) / g_depths.max() if g_depths.max() > 0 else np.nan
    if np.isnan(want):
        assert not np.isfinite(float(got))
    else:
        assert abs(float(got) - want) <= TOL
    assert F.equals(Fc)
    if relax:
        assert abs(float(got) - fx["reference_value"][0]) > 1e-6      # the quirk is NOT reproduced


@pytest.mark.parametrize("containment", ["simplex", "l1"])
@pytest.mark.parametrize("method", ["p1", "p2", "p3", "p4"])
def test_pointcloud_homogeneity_vs_restatement(oracle, method, containment):
    """PointcloudHomogeneity (homogeneity.py:155-200; the reference cannot run it on pandas >= 2): batched external-point
    launches against the definition restated with the oracle -- depth of g inside F u {g}, n_F + 1 points."""
    from statdepth_amd.homogeneity import PointcloudHomogeneity
    rng = np.random.default_rng(31)
    F = pd.DataFrame(rng.normal(size=(13, 2)))
    G = pd.DataFrame(rng.normal(size=(13, 2)) * 0.8 + 0.2)     # same labels as F: rows 0..12

    def depths(P):
        return oracle.pointcloud_depths(P) if containment == "simplex" else oracle.l1_depth(P)

    def inside(host, pt):
        Hp = np.concatenate([host, pt[None, :]])
        return (oracle.pointcloud_depths(Hp, [len(host)]) if containment == "simplex"
                else oracle.l1_depth(Hp, [len(host)]))[0]

    Fx, Gx = F.to_numpy(), G.to_numpy()
    dF, dG = depths(Fx), depths(Gx)
    assert (dF == dF.max()).sum() == 1 and (dG == dG.max()).sum() == 1      # a unique deepest point in each sample
    g_in_F = inside(Fx, Gx[np.argmax(dG)])
    p3 = max(inside(Fx, g) for g in Gx) / dG.max()
    want = {"p1": g_in_F / dF.max(), "p2": 1 - abs(g_in_F - dF.max()), "p3": p3,
            "p4": abs(p3 - inside(Fx, Fx[np.argmax(dF)]) / dF.max()) * abs(p3 - inside(Gx, Gx[np.argmax(dG)]) / dG.max())}[method]
    h = PointcloudHomogeneity(F, G, method=method, containment=containment)
    if np.isnan(want):          # l1, p4: a sample's own point appended to it coincides with itself -> 0/0, as in the reference
        assert method == "p4" and containment == "l1" and np.isnan(float(h.homogeneity()))
    else:
        assert abs(float(h.homogeneity()) - want) <= 1e-12 * max(1.0, abs(want))
    assert_depths_close(h.F_depths().to_numpy(), dF, TOL)
    assert_depths_close(h.G_depths().to_numpy(), dG, TOL)


def test_duplicate_column_labels_are_refused():
    """Decision pinned against _helper.py:32: with duplicated labels the reference merges curves inside its bands and
    raises TypeError for every target whose label is duplicated, i.e. for the default call (fixture g11_duplabels holds
    what it does).  The build refuses such frames with a ValueError before touching the device."""
    from statdepth_amd import FunctionalDepth
    fx = load_golden("g11_duplabels")
    df = frame_df(fx["input"])
    assert not df.columns.is_unique
    ref_default = [c for c in fx["calls"] if c["to_compute"] is None]
    assert all(c.get("raises") == "TypeError" for c in ref_default)
    for relax in (True, False):
        for tc in (None, ["a"], ["b"]):
            with pytest.raises(ValueError, match="unique"):
                FunctionalDepth([df], to_compute=tc, relax=relax)
    with pytest.raises(ValueError, match="unique"):
        FunctionalDepth([df], K=2, relax=True)


# ---------------------------------------------------------------- totals beyond int64
def _wide_reference(X, targets, J):
    """Python big-integer restatement of the closed form (no NaN): sum_t C(n-1,j) - C(A,j) - C(B,j)."""
    import math
    T, n = X.shape
    out = []
    for q in targets:
        x = X[:, q]
        A = (X > x[:, None]).sum(axis=1)
        B = (X < x[:, None]).sum(axis=1)
        out.append([sum(math.comb(n - 1, j) - math.comb(int(a), j) - math.comb(int(b), j) for a, b in zip(A, B))
                    for j in range(2, J + 1)])
    return out


def test_wide_totals_beyond_int64(eng, oracle):
    """sd_mbd_counts returns SD_ERR_OVERFLOW when T * C(n-1, J) >= 2^63; sd_mbd_counts_wide computes the totals as two
    64-bit limbs (timepoints in chunks that fit int64, added with carry), and FunctionalDepth falls through to it."""
    import math
    from statdepth_amd import FunctionalDepth
    from statdepth_amd._native import StatdepthHipError, SD_ERR_OVERFLOW
    rng = np.random.default_rng(19)
    T, n, J = 30, 3000, 6                                   # C(2999, 6) ~ 1.0e18: every chunk holds 9 timepoints
    X = np.round(rng.normal(size=(T, n)).cumsum(axis=0), 2)
    assert T * math.comb(n - 1, J) >= 2 ** 63 and J * math.comb(n - 1, J) < 2 ** 63
    tg = np.array([0, 7, 1500, 2999])
    for algo in ("pairwise", "rank"):
        with pytest.raises(StatdepthHipError) as ei:
            eng.mbd_counts(X, tg, J, algo=algo)
        assert ei.value.code == SD_ERR_OVERFLOW
        got = eng.mbd_counts_wide(X, tg, J, algo=algo)
        want = _wide_reference(X, tg, J)
        assert [[int(v) for v in row] for row in got] == want, algo
        assert max(max(row) for row in want) >= 2 ** 63     # the case really needs the second limb
        # curve-major input: the chunks are row ranges of either layout
        assert [[int(v) for v in row] for row in eng.mbd_counts_wide(np.asfortranarray(X), tg, J, algo=algo)] == want
    # where int64 suffices the two paths agree
    Y = X[:6, :200]
    small = eng.mbd_counts(Y, None, 3)
    wide = eng.mbd_counts_wide(Y, None, 3)
    assert (small == np.array([[int(v) for v in row] for row in wide], dtype=np.int64)).all()
    # through the API: depth = total / T / C(n, j) with the reference's normalisers
    df = pd.DataFrame(X)
    d = FunctionalDepth([df], to_compute=[0, 1500], J=J, relax=True)
    want_d = [sum(w[j - 2] / T / math.comb(n, j) for j in range(2, J + 1)) for w in _wide_reference(X, [0, 1500], J)]
    assert_depths_close(d.to_numpy(), np.array(want_d), TOL)


# ---------------------------------------------------------------- 'r2_enum': componentwise band containment
@pytest.mark.parametrize("d", [1, 2, 3, 5, 8])
def test_componentwise_band_vs_literal_enumeration(eng, oracle, d):
    """sd_multi_band_counts (pairs counted through the 3^d state classes) against the literal pair enumeration, ties
    included; d = 1 against the univariate kernels; strict form through K3 on the component series."""
    from statdepth_amd import FunctionalDepth
    rng = np.random.default_rng(300 + d)
    n, T = 23, 7
    P = np.round(rng.normal(size=(n, T, d)).cumsum(axis=1), 1)          # rounded: ties in single components
    P[3] = P[4]                                                          # a duplicated curve
    P[5, :, 0] = P[6, :, 0]
    want = oracle.multi_band_enum(P, None, 2, True)[:, 0]
    assert want.sum() > 0
    assert (eng.multi_band_counts(P) == want).all()
    tg = np.array([22, 0, 4])
    assert (eng.multi_band_counts(P, tg) == want[tg]).all()
    if d == 1:
        assert (eng.mbd_counts(np.ascontiguousarray(P[:, :, 0].T), None, 2)[:, 0] == want).all()
    frames = [pd.DataFrame(P[i]) for i in range(n)]
    import math
    got = FunctionalDepth(frames, containment="r2_enum", relax=True)
    assert_depths_close(got.to_numpy(), want / T / math.comb(n, 2), TOL)
    for J in (2, 3):
        strict = FunctionalDepth(frames, J=J, containment="r2_enum", relax=False, to_compute=[0, 3, 10])
        ws = oracle.multi_band_enum(P, [0, 3, 10], J, False)
        wd = sum(ws[:, j - 2] / math.comb(n, j) for j in range(2, J + 1))
        assert_depths_close(strict.to_numpy(), wd, TOL)
    with pytest.raises(ValueError, match="NaN"):
        Pn = P.copy(); Pn[1, 2, 0] = np.nan
        eng.multi_band_counts(Pn)


def test_config4_componentwise_band_full_size(eng, oracle):
    """BASELINE.json configs[3] in its componentwise-band form (SURVEY.md 8(a) M1 alternative (iii)): 5 000 curves x 500
    timepoints x 8 features, exact pair counts for every target; checked on a few targets against per-timepoint literal
    counting in numpy, plus the d = 1 / monotonicity properties at full size."""
    import torch
    n, T, d = 5000, 500, 8
    g = torch.Generator(device="cuda").manual_seed(1236)
    P = torch.randn(n, T, d, dtype=torch.float64, device="cuda", generator=g).cumsum(1)
    P[:16] *= 0.05
    got = eng.multi_band_counts(P)
    assert got.shape == (n,) and got.min() >= 0 and got.max() <= T * (n - 1) * (n - 2) // 2
    Ph = P.cpu().numpy()
    for q in (0, 7, 2500, 4999):
        tot = 0
        for t in range(0, T):
            up = (Ph[:, t, :] > Ph[q, t, :]).astype(np.uint16)
            dn = (Ph[:, t, :] < Ph[q, t, :]).astype(np.uint16)
            w = (1 << np.arange(d)).astype(np.uint16)
            mask = (up @ w) | ((dn @ w) << 8)
            mask = np.delete(mask, q)
            cnt = np.bincount(mask, minlength=65536)
            vals = np.nonzero(cnt)[0]
            disj = (vals[:, None] & vals[None, :]) == 0
            ordered = int((cnt[vals][:, None] * cnt[vals][None, :] * disj).sum())
            tot += (ordered - int(cnt[0])) // 2
        assert int(got[q]) == tot, q
    assert got[:16].mean() > 10 * max(1.0, got[16:].mean())     # the central curves are deep, the rest much less
    one = eng.multi_band_counts(P[:, :, :1].contiguous())
    assert (one >= got).all()                                    # fewer features: never fewer containing pairs


def test_homogeneity_with_block_sampling_is_reproducible(oracle):
    """The K-sampled callers of the homogeneity coefficients (curves and point clouds) consume the global numpy RNG like
    the reference's estimators do: the same seed gives the same coefficient, every value is finite, and the caller's
    frames stay untouched."""
    from statdepth_amd.homogeneity import FunctionalHomogeneity, PointcloudHomogeneity
    rng = np.random.default_rng(61)
    F = pd.DataFrame(rng.normal(size=(12, 10)).cumsum(axis=0), columns=[f"F{i}" for i in range(10)])
    G = pd.DataFrame(rng.normal(size=(12, 8)).cumsum(axis=0) + 0.3, columns=[f"G{i}" for i in range(8)])
    Fc, Gc = F.copy(), G.copy()
    vals = []
    for _ in range(2):
        np.random.seed(314)
        vals.append(float(np.asarray(FunctionalHomogeneity([F], [G], method="p3", K=2, relax=True, quiet=True).homogeneity()).ravel()[0]))
    assert vals[0] == vals[1] and np.isfinite(vals[0])
    assert F.equals(Fc) and G.equals(Gc)
    P = pd.DataFrame(rng.normal(size=(14, 2)))
    Q = pd.DataFrame(rng.normal(size=(14, 2)) + 0.2)
    pv = []
    for _ in range(2):
        np.random.seed(2718)
        pv.append(float(PointcloudHomogeneity(P, Q, method="p3", K=2).homogeneity()))
    assert pv[0] == pv[1] and np.isfinite(pv[0])
