#!/usr/bin/env python3
"""Differential fuzz on the GPU box: the product library's kernels against the predecessors kept in the cross-check
library (libstatdepth_hip_xcheck.so): second-generation strict kernels against the first generation, register-resident
simplex test against the generic one, large-n route variants.  usage: fuzz_secondary.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()                                             # the runtime is up before a second HIP library is opened
from statdepth_amd import engine, _native
PRODUCT = _native.load()
XCHECK = _native.open_library(_native.XCHECK_LIB_PATH)      # the switches below are honoured by this build only

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
def env(k, v):
    if v is None:
        os.environ.pop(k, None)
        if not any(e.startswith("SD_") for e in os.environ): _native._LIB = PRODUCT
    else:
        os.environ[k] = v
        _native._LIB = XCHECK
for c in range(cases):
    # ---- strict band depth ----
    n = int(rng.choice([rng.integers(3, 70), rng.integers(70, 600), rng.integers(600, 1500)]))
    T = int(rng.choice([rng.integers(1, 40), rng.integers(40, 200), rng.integers(900, 1100)]))
    kind = rng.choice(["bands", "walk", "ints", "normal"])
    if kind == "bands": X = np.round(np.sort(rng.normal(size=n))[None, :] * 3 + rng.normal(size=(T, n)) * 0.3, 1)
    elif kind == "walk": X = rng.normal(size=(T, n)).cumsum(axis=0)
    elif kind == "ints": X = rng.integers(-2, 3, size=(T, n)).astype(float)
    else: X = rng.normal(size=(T, n))
    if rng.random() < 0.3: X[rng.random(X.shape) < 0.01] = np.nan
    if rng.random() < 0.3: X[:, rng.integers(0, n)] = X[:, rng.integers(0, n)]
    m = min(n, 64)
    tg = np.sort(rng.choice(n, size=m, replace=False))
    env("SD_STRICT_V1", None); a = engine.bd_strict_counts(X, tg, 2)
    env("SD_STRICT_V1", "1"); b = engine.bd_strict_counts(X, tg, 2)
    env("SD_STRICT_V1", None)
    if not (a == b).all():
        bad += 1; print(f"STRICT MISMATCH case {c}: n={n} T={T} kind={kind}", flush=True)
    # ---- simplex (pointcloud, exhaustive small / sampled) ----
    d = int(rng.integers(1, 9))
    npts = int(rng.integers(d + 2, d + 9))
    P = rng.normal(size=(npts, d))
    if rng.random() < 0.3: P = np.round(P, 0)                     # degenerate simplices
    if rng.random() < 0.2: P[rng.integers(0, npts)] = P[rng.integers(0, npts)]
    env("SD_SIMPLEX_GENERIC", None); a = engine.pointcloud_simplex_counts(P)
    env("SD_SIMPLEX_GENERIC", "1"); b = engine.pointcloud_simplex_counts(P)
    env("SD_SIMPLEX_GENERIC", None)
    if not (a == b).all():
        bad += 1; print(f"SIMPLEX MISMATCH case {c}: n={npts} d={d}", flush=True)
    Q = rng.normal(size=(40, int(rng.integers(2, 6)), d))
    env("SD_SIMPLEX_GENERIC", None); a = engine.multi_simplex_counts(Q, samples=64, seed=c)
    env("SD_SIMPLEX_GENERIC", "1"); b = engine.multi_simplex_counts(Q, samples=64, seed=c)
    env("SD_SIMPLEX_GENERIC", None)
    if not (a == b).all():
        bad += 1; print(f"MULTI SIMPLEX MISMATCH case {c}: d={d}", flush=True)
    # ---- large-n route: sort-free ranking + staged partition against the predecessors ----
    if c % 5 == 0:
        n2 = int(rng.integers(16385, 60000)); T2 = int(rng.integers(1, 6))
        Y = rng.normal(size=(T2, n2)).cumsum(axis=0)
        if rng.random() < 0.5: Y = np.round(Y, 1)
        if rng.random() < 0.3: Y[rng.random(Y.shape) < 0.001] = np.nan
        tg2 = np.sort(rng.choice(n2, size=300, replace=False))
        a = engine.mbd_counts(Y, tg2, 2, algo="rank")
        env("SD_BIG_SORT", "1"); env("SD_BIG_PART1", "1")
        b = engine.mbd_counts(Y, tg2, 2, algo="rank")
        env("SD_BIG_SORT", None); env("SD_BIG_PART1", None)
        if not (a == b).all():
            bad += 1; print(f"LARGE-N MISMATCH case {c}: n={n2} T={T2}", flush=True)
    if c % 20 == 19: print(f"{c + 1} cases, {bad} mismatches", flush=True)
print("FUZZ OK" if bad == 0 else f"FUZZ FAILED: {bad}")
sys.exit(1 if bad else 0)
