"""CPU-side tests: the C-ABI library loads and exports what include/statdepth_hip.h declares,
the host mirror of the reference's API validates arguments the same way, result objects
behave like the reference's, and the product fails loudly without a GPU (no CPU fallback)."""
import os
import sys
import re

import numpy as np
import pandas as pd
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from statdepth_amd import _native
    return _native


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "statdepth_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sd_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built):
    import ctypes
    lib = ctypes.CDLL(built.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in statdepth_hip.h but not exported"
    # and the Python binding table covers exactly the header
    assert sorted(built.SIGNATURES) == declared


def test_product_library_has_no_crosscheck_code(built):
    """The product library is built without -DSD_CROSSCHECK: it says so, it does not contain the retired kernel
    generations, and the package never opens the cross-check build (tests do, through conftest.xcheck)."""
    import ctypes
    import subprocess
    prod = ctypes.CDLL(built.LIB_PATH)
    xchk = ctypes.CDLL(built.XCHECK_LIB_PATH)
    assert prod.sd_is_crosscheck_build() == 0 and xchk.sd_is_crosscheck_build() == 1
    for name in _declared_symbols():
        assert hasattr(xchk, name)
    retired = ("rank_packed_kernel", "rank_search_kernel", "launch_mbd_rank_v1", "bucket_packed_kernel",
               "bucket_partition_kernel")
    syms = {p: subprocess.run(["nm", "-C", p], capture_output=True, text=True).stdout for p in (built.LIB_PATH, built.XCHECK_LIB_PATH)}
    for r in retired:
        assert not re.search(rf"\b{r}\b", syms[built.LIB_PATH]), f"{r} is in the product library"
        assert re.search(rf"\b{r}\b", syms[built.XCHECK_LIB_PATH]), f"{r} missing from the cross-check library"
    assert "getenv" not in subprocess.run(["nm", "-D", "-u", built.LIB_PATH], capture_output=True, text=True).stdout, \
        "the product library reads no environment variable"
    pkg = ""
    for dp, _, fns in os.walk(os.path.join(ROOT, "statdepth_amd")):
        for fn in fns:
            if fn.endswith(".py") and fn != "_native.py":
                pkg += open(os.path.join(dp, fn)).read()
    assert "XCHECK_LIB_PATH" not in pkg and "xcheck" not in pkg


def test_abi_version_and_error_string(built):
    lib = built.load()
    assert lib.sd_abi_version() == 1
    # argument validation happens before any device work: callable without a GPU
    rc = lib.sd_mbd_counts(None, 10, 10, 10, 1, None, 10, 2, 0, None, None, 0, None)
    assert rc == built.SD_ERR_INVALID
    assert b"null" in lib.sd_last_error()
    rc = lib.sd_mbd_counts(1, 10, 10, 3, 7, None, 10, 2, 0, 1, None, 0, None)
    assert rc == built.SD_ERR_INVALID and b"time-major" in lib.sd_last_error()
    rc = lib.sd_mbd_counts(1, 1000, 100000, 100000, 1, None, 100000, 4, 0, 1, None, 0, None)
    assert rc == built.SD_ERR_OVERFLOW
    rc = lib.sd_bd_strict_j_counts(1, 10, 10, 10, 1, None, 10, 5, 1, None, 0, None)
    assert rc == built.SD_ERR_UNSUPPORTED
    rc = lib.sd_pointcloud_simplex_counts(1, 10, 9, None, 10, 1e-7, 1, None)
    assert rc == built.SD_ERR_UNSUPPORTED
    assert lib.sd_mbd_workspace_bytes(1000, 10000, 10000, 1, 10000, 2, 0) >= 1000 * 4


def test_no_cpu_fallback(built):
    """Without a device the product raises; it never routes to the oracle or any CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from statdepth_amd import FunctionalDepth, PointcloudDepth
    df = pd.DataFrame(np.random.default_rng(0).normal(size=(6, 5)))
    with pytest.raises(RuntimeError, match="no HIP device"):
        FunctionalDepth([df], relax=True)
    with pytest.raises(RuntimeError, match="no HIP device"):
        FunctionalDepth([df], relax=False)
    with pytest.raises(RuntimeError, match="no HIP device"):
        PointcloudDepth(df.iloc[:, :2], containment="l1")
    src = ""
    for dp, _, fns in os.walk(os.path.join(ROOT, "statdepth_amd")):
        for fn in fns:
            if fn.endswith(".py"):
                src += open(os.path.join(dp, fn)).read()
    assert "import oracle" not in src and "from oracle" not in src


def test_handle_depth_errors_mirror_reference():
    """Same checks, same order, same exception types as _helper.py:59-107."""
    from statdepth_amd import DepthDegeneracy, FunctionalDepth
    df = pd.DataFrame(np.arange(20.0).reshape(5, 4))
    with pytest.raises(ValueError, match="passed as a list"):
        FunctionalDepth(df)
    with pytest.raises(ValueError, match="J must be an integer"):
        FunctionalDepth([df], J=2.0)
    with pytest.raises(ValueError, match="str or Callable"):
        FunctionalDepth([df], containment=3)
    with pytest.raises(ValueError, match="deep_check"):
        FunctionalDepth([df], deep_check=1)
    with pytest.raises(ValueError, match="relax must be"):
        FunctionalDepth([df], relax=1)
    with pytest.raises(ValueError, match="greater than or equal to 2"):
        FunctionalDepth([df], J=1)
    with pytest.raises(ValueError, match="No data"):
        FunctionalDepth([])
    with pytest.raises(ValueError, match="less than the number of observations"):
        FunctionalDepth([df], J=5)          # J compared with the number of ROWS (timepoints), _helper.py:83
    with pytest.raises(ValueError, match="invalid for multivariate"):
        FunctionalDepth([df, df, df], containment="r2")
    with pytest.raises(DepthDegeneracy):
        FunctionalDepth([df, df, df], containment="simplex")     # needs >= d + 2 = 6 functions
    with pytest.raises(ValueError, match="is invalid"):
        FunctionalDepth([df], containment="nonsense")
    with pytest.raises(ValueError, match="incorrect number of parameters"):
        FunctionalDepth([df], containment=lambda a, b: 0.0)
    with pytest.raises(NotImplementedError):
        FunctionalDepth([df], containment="r2_enum")
    obj = pd.DataFrame({"a": ["x", "y", "z"], "b": [1, 2, 3]})
    with pytest.raises(ValueError, match="numeric dtypes"):
        FunctionalDepth([obj], deep_check=True)
    from statdepth_amd import PointcloudDepth
    with pytest.raises(ValueError, match="not a valid containment"):
        PointcloudDepth(df, containment="nonsense")
    with pytest.raises(DepthDegeneracy, match="Block size"):
        FunctionalDepth([df], K=9)


def test_custom_containment_callable_runs_on_host():
    """The plug-in protocol (docs/index.md:124-148) is host logic: one call per subset, keyword arguments."""
    from statdepth_amd import FunctionalDepth
    df = pd.DataFrame({"a": [1.0, 2, 3], "b": [2.0, 3, 4], "c": [0.0, 5, 1], "d": [1.5, 2.5, 3.5]})
    calls = []

    def inside(data, curve, relax):
        calls.append((tuple(data.columns), curve.name, relax))
        ok = ((data.min(axis=1) <= curve) & (curve <= data.max(axis=1))).sum()
        return ok / len(curve) if relax else ok // len(curve)

    got = FunctionalDepth([df], containment=inside, relax=True)
    assert len(calls) == 4 * 3 and calls[0][2] is True
    # same numbers as the built-in definition (checked against the oracle, no GPU involved)
    import oracle
    want = oracle.univariate_depths(df.to_numpy(), None, J=2, relax=True)
    assert np.allclose(got.to_numpy(), want, atol=1e-12)
    assert list(got.index) == list(df.columns)


def test_result_objects_behave_like_reference():
    from statdepth_amd.depth.depth import _FunctionalDepthUnivariate, _PointwiseDepth
    df = pd.DataFrame(np.arange(12.0).reshape(3, 4), columns=list("wxyz"))
    depths = pd.Series([0.2, 0.4, 0.1, 0.3], index=list("wxyz"))
    r = _FunctionalDepthUnivariate(df=df, depths=depths)
    assert isinstance(r, pd.Series) and r.get_data() is df
    assert list(r.ordered().index) == ["x", "z", "w", "y"]
    assert list(r.deepest(n=2).index) == ["x", "z"] and r.median().index[0] == "x"
    assert list(r.outlying(n=2).index) == ["w", "y"] and r.outlying().index[0] == "y"
    assert list(r.sorted().index) == ["x", "z", "w", "y"]
    assert list(r.quartile().index) == ["y", "w"]            # lower half, `ratio` ignored like the reference
    assert r.depths() is depths and r.get_depths() is depths
    assert list(r.drop_outlying_data(n=1).columns) == ["w", "x", "z"]
    assert list(r.get_deepest_data(n=2).columns) == ["x", "z"]
    assert list(r.get_outlying_data(n=1).columns) == ["y"]
    pts = pd.DataFrame(np.arange(8.0).reshape(4, 2), index=list("abcd"))
    p = _PointwiseDepth(df=pts, depths=pd.Series([0.1, 0.5, 0.3, 0.2], index=list("abcd")))
    assert list(p.drop_outlying_data(n=1).index) == ["b", "c", "d"]
    assert list(p.get_deepest_data(n=2).index) == ["b", "c"]


def test_device_matrix_accepts_both_pandas_layouts():
    """Ingest keeps the frame's memory order: strides are what the C ABI receives."""
    a = np.arange(12.0).reshape(3, 4)
    c = pd.DataFrame(a)                                   # C-contiguous T x n
    f = pd.DataFrame({k: a[:, k] for k in range(4)})      # column-built
    xc = c.to_numpy(dtype=np.float64, copy=False)
    xf = f.to_numpy(dtype=np.float64, copy=False)
    assert xc.flags.c_contiguous and (xc == xf).all()


def test_plot_helpers_build_figures():
    """The plotly views of the result objects (reference depth.py:91-175,193-335): figures come back with the expected
    traces -- everything plain, the deepest / most outlying items marked -- without any device work."""
    import plotly.graph_objects as go
    from statdepth_amd.depth.depth import _FunctionalDepthUnivariate, _PointwiseDepth
    rng = np.random.default_rng(1)
    df = pd.DataFrame(rng.normal(size=(9, 6)), columns=list("abcdef"))
    res = _FunctionalDepthUnivariate(df=df, depths=pd.Series(index=df.columns, data=[.1, .5, .3, .2, .4, .0]))
    fig = res.plot_deepest(n=2, title="t", return_plot=True)
    assert isinstance(fig, go.Figure) and len(fig.data) == 6
    assert [tr.name for tr in fig.data[-2:]] == ["b", "e"] and all(tr.line.color == "Red" for tr in fig.data[-2:])
    fig = res.plot_outlying(n=1, return_plot=True, showlegend=True)
    assert fig.data[-1].name == "f" and fig.layout.showlegend is True
    for d in (2, 3):
        pc = pd.DataFrame(rng.normal(size=(12, d)))
        pres = _PointwiseDepth(df=pc, depths=pd.Series(index=pc.index, data=rng.random(12)))
        assert len(pres.plot_deepest(n=3, return_plot=True).data) == 2
        assert len(pres.plot_outlying(n=3, return_plot=True).data[1].x) == 3
        assert len(pres.plot_depths(invert_colors=True, return_plot=True).data) == 1
    with pytest.raises(ValueError, match="Dimensionality"):
        _PointwiseDepth(df=pc.iloc[:, :1], depths=pd.Series(index=pc.index, data=rng.random(12))).plot_depths(return_plot=True)


def test_bench_contract_and_committed_evidence():
    """bench.py's helpers without a GPU: the PMC traffic it quotes comes from a committed profile of the same workload and
    is labelled with its file, and the round's recorded bench line carries the objects the measurement contract names."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    traffic, source = bench.pmc_traffic(["rank_bucket_kernel", "rank_finalize"], 10000, 1000, 2)
    assert source and source.startswith("profiles/") and os.path.exists(os.path.join(ROOT, source))
    assert 0.5 * 160.08e6 < traffic < 1.5 * 160.08e6
    assert bench.pmc_traffic(["rank_bucket_kernel"], 9999, 1000, 2) == (None, None)      # another workload: no quote
    lines = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if "bench_line" in f and f.endswith(".json"))
    assert lines
    with open(os.path.join(ROOT, "profiles", lines[-1])) as f:
        line = json.load(f)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert line["dtype"] == "f64" and line["vs_baseline"] is None and "workload" in line["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in line["roofline"], key
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-9


def test_committed_profiles_are_of_the_shipped_kernels():
    """The newest committed kernel statistics of the headline workload must not predate the last commit that touched the
    kernels it measures: a stale profile is evidence of another build."""
    import subprocess

    def last_commit_time(paths):
        out = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%ct", "--"] + paths, capture_output=True, text=True)
        return int(out.stdout.strip()) if out.returncode == 0 and out.stdout.strip() else None

    if last_commit_time(["bench.py"]) is None:
        pytest.skip("no git history here")
    stats = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("bench_kernel_stats.csv"))
    assert stats
    newest = os.path.join("profiles", stats[-1])
    with open(os.path.join(ROOT, newest)) as f:
        names = f.read()
    assert "rank_bucket32_kernel" in names, "the committed headline profile is not of the two-launch path"
    t_prof = last_commit_time([newest])
    t_kern = last_commit_time(["statdepth_amd/csrc/mbd_rank_bucket32.hip", "statdepth_amd/csrc/mbd_rank_bucket.hip",
                               "statdepth_amd/csrc/rank_bucket.h"])
    if t_prof is None:
        pytest.skip("profile not committed yet")
    assert t_prof >= t_kern, f"{newest} predates the last change of the kernels it measures: run tools/collect_profiles.py"
    # bench.py quotes traffic and kernel time from the newest files that hold the kernels of the step
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    traffic, source = bench.pmc_traffic(["rank_bucket32_kernel", "rank_bucket_kernel"], 10000, 1000, 2)
    assert source and 0.5 * 160.08e6 < traffic < 1.5 * 160.08e6
    kus, ksrc = bench.profiled_kernel_us("rank_bucket32_kernel", 10000, 1000, 2)
    assert ksrc == newest and 20.0 < kus < 80.0
    # ... and the strict leg's issue roofline from SQ counters collected on the shipped strict kernels (a
    # constant copied from an earlier round's file is not a measurement of this build)
    iss = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("issue_strict.json"))
    assert iss
    t_iss = last_commit_time([os.path.join("profiles", iss[-1])])
    t_strict = last_commit_time(["statdepth_amd/csrc/bd_strict.hip"])
    if t_iss is not None:
        assert t_iss >= t_strict, f"profiles/{iss[-1]} predates the last change of bd_strict.hip: run tools/issue_strict.py"
    roof = bench.strict_roofline()
    assert roof and roof["counter_file"] == os.path.join("profiles", iss[-1])


def test_bench_self_launch_plan():
    """`python bench.py --gpus N` (N > 1) without a launcher starts its own ranks from a parent that never touches the GPU;
    under a launcher (WORLD_SIZE set) or at N = 1 the process is a rank itself.  Dry run only: no rank is started here."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    bench = os.path.join(ROOT, "bench.py")

    def plan(args, **extra):
        out = subprocess.run([sys.executable, bench] + args + ["--print-launch"], env={**env, **extra}, capture_output=True,
                             text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        return json.loads(out.stdout.strip().splitlines()[-1])

    p = plan(["--gpus", "4", "--steps", "7", "--warmup", "2"])
    assert p["self_launch"] and p["world_size"] == 4
    a = p["argv"]
    assert a[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in a and "--nnodes=1" in a
    assert a[a.index("--master-addr") + 1] == "127.0.0.1" and a[a.index("--master-port") + 1].isdigit()
    k = a.index(bench)
    assert a[k + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]          # the children run the same command line
    assert not plan([])["self_launch"] and not plan(["--gpus", "1"])["self_launch"]
    under = plan(["--gpus", "4"], WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    assert not under["self_launch"] and under["world_size"] == 4
    # the parent of a self-launched run must not have imported torch (and so cannot have initialised HIP)
    src = open(bench).read()
    head = src[:src.index("def main():")]
    assert "import torch" not in head
