import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _dec(v):
    return float(v) if isinstance(v, str) else v


def load_golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def golden_names(kind=None, prefix=None):
    out = []
    for fn in sorted(os.listdir(GOLDEN)):
        if not fn.endswith(".json"):
            continue
        nm = fn[:-5]
        if prefix and not nm.startswith(prefix):
            continue
        if kind:
            with open(os.path.join(GOLDEN, fn)) as f:
                if json.load(f).get("kind") != kind:
                    continue
        out.append(nm)
    return out


def frame_values(fj):
    """fixture frame -> float64 ndarray (rows x cols), specials decoded."""
    return np.array([[_dec(v) for v in row] for row in fj["values"]], dtype=np.float64)


def frame_df(fj):
    import pandas as pd
    return pd.DataFrame(frame_values(fj), index=fj["index"], columns=fj["columns"])


def depths_of(fx):
    return np.array([_dec(v) for v in fx["depths"]], dtype=np.float64)


def assert_depths_close(got, want, tol=1e-12):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape
    nan_g, nan_w = np.isnan(got), np.isnan(want)
    assert (nan_g == nan_w).all()
    ok = ~nan_w
    assert np.all(np.abs(got[ok] - want[ok]) <= tol * np.maximum(1.0, np.abs(want[ok]))), (got, want)


@pytest.fixture(scope="session")
def oracle():
    import oracle as o
    o.build()
    return o


@pytest.fixture
def xcheck(monkeypatch):
    """Cross-check implementations live in libstatdepth_hip_xcheck.so (the -DSD_CROSSCHECK build of the same sources: the
    retired kernel generations + the environment switches that select them); the product library has neither.

        with xcheck(SD_RANK_IMPL="3"):
            other = engine.mbd_counts(...)      # runs on the cross-check library with the switch set

    Outside the `with` the engine is back on the product library."""
    import contextlib
    from statdepth_amd import _native
    state = {}

    @contextlib.contextmanager
    def use(**env):
        if "lib" not in state:
            state["lib"] = _native.open_library(_native.XCHECK_LIB_PATH)
            assert state["lib"].sd_is_crosscheck_build() == 1
        product = _native.load()
        with monkeypatch.context() as m:
            for k, v in env.items():
                m.setenv(k, str(v))
            m.setattr(_native, "_LIB", state["lib"])
            yield state["lib"]
        assert _native._LIB is product
    return use
