#!/usr/bin/env python3
"""Differential fuzz of round 3's strict-depth paths on the GPU box: the product library against the cross-check library's
independent implementations -- the pair kernel that tests every partner of a dirty curve (SD_STRICT_PAIRS2), the fp64 mask
kernel (SD_STRICT_FP64_MASKS, also the switch that takes n > 32 767 off the 32-bit rank image) and, on small cases, the first
generation (SD_STRICT_V1: every pair tested).  Shapes: n from a few hundred to 45 000 (LDS table + small filter, table area as
the filter above 13 107 curves, 32-bit ranks above 32 767), tie-heavy / banded / walks / integers, NaN, duplicated curves,
constant timepoints; short series (T <= 8: state classes -- lane and workgroup forms -- against SD_STRICT_NOCLASS).  usage: fuzz_strict3.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from statdepth_amd import engine, _native
PRODUCT = _native.load()
XCHECK = _native.open_library(_native.XCHECK_LIB_PATH)

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0


def run(X, tg, **env):
    for k, v in env.items():
        os.environ[k] = v
    _native._LIB = XCHECK if env else PRODUCT
    try:
        return engine.bd_strict_counts(X, tg, 2)[:, 0]
    finally:
        for k in env:
            os.environ.pop(k, None)
        _native._LIB = PRODUCT


for c in range(cases):
    cls = rng.choice(["short", "small", "mid", "big", "huge"], p=[0.2, 0.25, 0.25, 0.15, 0.15])
    if cls == "short": n, T = int(rng.choice([rng.integers(3, 400), rng.integers(400, 20000)])), int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8]))
    elif cls == "small": n, T = int(rng.integers(50, 900)), int(rng.choice([33, 64, 100, 257, 1000]))
    elif cls == "mid": n, T = int(rng.integers(900, 6000)), int(rng.choice([40, 96, 130, 300]))
    elif cls == "big": n, T = int(rng.integers(13200, 30000)), int(rng.choice([33, 64, 70]))
    else: n, T = int(rng.integers(32800, 45000)), int(rng.choice([33, 40, 64]))
    kind = rng.choice(["banded", "banded_rounded", "walks", "walks_rounded", "ints", "mirror"])
    lev = np.sort(rng.normal(size=n))[None, :] * 3.0
    if kind == "banded": X = lev + rng.normal(size=(T, n)) * 0.3
    elif kind == "banded_rounded": X = np.round(lev + rng.normal(size=(T, n)) * 0.3, int(rng.choice([0, 1, 2])))
    elif kind == "walks": X = rng.normal(size=(T, n)).cumsum(axis=0)
    elif kind == "walks_rounded": X = np.round(rng.normal(size=(T, n)).cumsum(axis=0), int(rng.choice([0, 1])))
    elif kind == "ints": X = (np.sort(rng.integers(0, 60, size=n))[None, :] + rng.integers(-1, 2, size=(T, n))).astype(float)
    else:                                                       # mirrored curves: many pairs of complementary masks
        h = n // 2
        base = rng.normal(size=(T, h)).cumsum(axis=0)
        X = np.concatenate([base, -base, rng.normal(size=(T, n - 2 * h))], axis=1)
    if rng.random() < 0.3: X[rng.integers(0, T, 30), rng.integers(0, n, 30)] = np.nan
    if rng.random() < 0.3: X[:, rng.integers(0, n)] = X[:, rng.integers(0, n)]
    if rng.random() < 0.2: X[rng.integers(0, T)] = 1.5          # a timepoint every curve shares
    if rng.random() < 0.2: X[0] = 0.0                           # a common start
    m = 48 if n > 6000 else min(n, 160)
    tg = np.sort(rng.choice(n, size=m, replace=False))
    a = run(X, tg)
    if cls == "short":                                          # state classes against masks + matching
        refs = {"NOCLASS": run(X, tg, SD_STRICT_NOCLASS="1")}
        if T in (4, 5): refs["LANECLASS"] = run(X, tg, SD_STRICT_LANECLASS="1")   # histogram per lane against the workgroup form
    else:
        refs = {"PAIRS2": run(X, tg, SD_STRICT_PAIRS2="1"), "FP64_MASKS": run(X, tg, SD_STRICT_FP64_MASKS="1")}
    if n <= 900 and T <= 300 and cls != "short":
        refs["V1"] = run(X, tg, SD_STRICT_V1="1")
    for name, b in refs.items():
        if not (a == b).all():
            bad += 1
            print(f"MISMATCH case {c} vs {name}: n={n} T={T} kind={kind} targets {tg[np.nonzero(a != b)[0][:5]]}", flush=True)
    if c % 10 == 9: print(f"{c + 1} cases, {bad} mismatches", flush=True)
print("FUZZ OK" if bad == 0 else f"FUZZ FAILED: {bad}")
sys.exit(1 if bad else 0)
