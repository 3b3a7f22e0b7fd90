"""Config-2-shaped test matrices of the timing tools, selected by environment switches (SD_TIES, SD_SORTED, SD_OUTLIER,
SD_OUTLIER_RANDOM, SD_CAUCHY; default: random walks)."""
import os
import numpy as np


def rank_data(n, T):
    X = np.random.default_rng(1234).normal(size=(T, n)).cumsum(axis=0)
    if os.environ.get("SD_TIES"):
        X = np.round(X, 1)
    if os.environ.get("SD_SORTED"):                  # curves ordered by level (normal / t3): the waves' subsets are stratified
        g = np.random.default_rng(3)
        lev = np.sort(g.normal(size=n) if os.environ["SD_SORTED"] == "normal" else g.standard_t(3, size=n))
        X = lev[None, :] * 30.0 + X * 0.05
    if os.environ.get("SD_OUTLIER"):                 # a few curves far outside the others' range
        X[:, :int(os.environ["SD_OUTLIER"])] *= float(os.environ.get("SD_OUTLIER_SCALE", "1e6"))
    if os.environ.get("SD_OUTLIER_RANDOM"):          # ... at random positions (every wave of the bucket kernel gets some)
        X[:, np.random.default_rng(9).choice(n, size=int(os.environ["SD_OUTLIER_RANDOM"]), replace=False)] *= 1e6
    if os.environ.get("SD_CAUCHY"):                  # heavy tails at every timepoint
        X = np.random.default_rng(5).standard_cauchy(size=(T, n))
    if os.environ.get("SD_NANROWS"):                 # a NaN in each of that many rows (rows left to the second launch)
        g = np.random.default_rng(11)
        for r in g.choice(T, size=int(os.environ["SD_NANROWS"]), replace=False):
            X[r, g.integers(0, n)] = np.nan
    return X
