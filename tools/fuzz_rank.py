#!/usr/bin/env python3
"""Differential fuzz on the GPU box: rank path (bucket kernel / large-n route) against the pairwise kernel (independent
code) on random shapes, distributions and specials; J = 2 and 3.  usage: fuzz_rank.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from statdepth_amd import engine

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(cases):
    n = int(rng.choice([rng.integers(2, 70), rng.integers(70, 1100), rng.integers(1100, 9000), rng.integers(9000, 16385),
                        rng.integers(16385, 45000)]))
    T = int(rng.integers(1, 12))
    if 16384 < n <= 40960 and rng.random() < 0.7:
        T = int(rng.integers(96, 104))                        # the medium route (column blocks) takes over from 96 rows on
    if rng.random() < 0.12:
        T = int(rng.integers(260, 640))                       # more rows than workgroups: the bracket-with-the-range mode of the map
    kind = rng.choice(["normal", "walk", "ints", "round", "cauchy", "const", "lognormal", "tiny", "huge", "mixed"])
    X = rng.normal(size=(T, n))
    if kind == "walk": X = X.cumsum(axis=0)
    elif kind == "ints": X = rng.integers(-3, 4, size=(T, n)).astype(float)
    elif kind == "round": X = np.round(X * rng.choice([1, 10, 100]), 0)
    elif kind == "cauchy": X = rng.standard_cauchy(size=(T, n))
    elif kind == "const": X[:] = rng.normal()
    elif kind == "lognormal": X = np.exp(X * 5)
    elif kind == "tiny": X = X * 1e-312
    elif kind == "huge": X = X * 1e307
    elif kind == "mixed":                                     # heavy-tailed and Gaussian stretches of rows taking turns
        hv = (np.arange(T) // int(rng.integers(1, 120))) % 2 == 1
        X[hv] = rng.standard_cauchy(size=(int(hv.sum()), n)) * rng.choice([1.0, 1e-3, 1e5])
    # outlying curves / entries (robust range), values a hair apart (mixed buckets), blocks of equal values inside
    # continuous rows (tie path with and without crowding)
    if rng.random() < 0.35: X[:, rng.choice(n, size=min(n, int(rng.choice([1, 3, 17, max(1, n // 50)]))), replace=False)] *= rng.choice([1e3, 1e6, 1e12])
    if rng.random() < 0.2: X[rng.random(X.shape) < 0.002] *= -1e8
    if rng.random() < 0.25:
        w = int(rng.integers(2, max(3, min(n, 400))))
        c0 = int(rng.integers(0, max(1, n - w)))
        v = X[:, c0:c0 + 1].copy()
        X[:, c0:c0 + w] = np.where(rng.random((T, min(w, n - c0))) < 0.5, v, v * (1 + rng.choice([0.0, 1e-15, 1e-12])))
    if rng.random() < 0.4: X[rng.random(X.shape) < rng.choice([0.0005, 0.02, 0.5])] = np.nan
    if rng.random() < 0.3: X[rng.random(X.shape) < 0.001] = np.inf
    if rng.random() < 0.3: X[rng.random(X.shape) < 0.001] = -np.inf
    if rng.random() < 0.2 and n > 3: X[:, rng.integers(0, n)] = X[:, rng.integers(0, n)]
    m = min(n, 400)
    tg = np.sort(rng.choice(n, size=m, replace=False))
    for J in (2, 3):
        if n - 1 < J: continue
        a = engine.mbd_counts(X, tg, J, algo="rank")
        b = engine.mbd_counts(X, tg, J, algo="pairwise")
        if not (a == b).all():
            bad += 1
            print(f"MISMATCH case {c}: n={n} T={T} kind={kind} J={J} first bad target {tg[np.nonzero((a != b).any(axis=1))[0][:5]]}", flush=True)
    if c % 20 == 19: print(f"{c + 1} cases, {bad} mismatches", flush=True)
print("FUZZ OK" if bad == 0 else f"FUZZ FAILED: {bad}")
sys.exit(1 if bad else 0)
