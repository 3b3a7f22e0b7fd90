#!/usr/bin/env python3
"""Differential fuzz of the two-launch bucket path (rank_bucket32_kernel + rank_bucket_kernel's SEL form; 3072 < n <= 11264,
256 <= T <= 4096, every curve a target) against the pairwise kernel (independent code) on the GPU box:
continuous rows, image collisions (values a hair apart), equal values, duplicated curves, quantised rows, NaN / inf rows,
outlying curves, constant rows.  usage: fuzz_rb32.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from statdepth_amd import engine

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(cases):
    n = int(rng.integers(3073, 11265))
    T = int(rng.choice([256, 257, 300, 511, 512, 513, 700, 1024]))
    kind = rng.choice(["walk", "normal", "round1", "round2", "ints", "lognormal", "mixed", "cauchy", "t3", "levels", "spikes", "tiny",
                       "huge", "clusters", "stride", "edge"])
    X = rng.normal(size=(T, n))
    if kind == "walk": X = X.cumsum(axis=0)
    elif kind == "round1": X = np.round(X.cumsum(axis=0), 1)
    elif kind == "round2": X = np.round(X * 100, 0) / 100
    elif kind == "ints": X = rng.integers(-40, 41, size=(T, n)).astype(float)
    elif kind == "lognormal": X = np.exp(X * 3)
    elif kind == "mixed":
        X = X.cumsum(axis=0)
        X[::3] = np.round(X[::3], 1)                                  # quantised rows between continuous ones
        X[1::7, : n // 2] = np.round(X[1::7, : n // 2], 0)            # half a row quantised: ties mixed with distinct values
    elif kind == "cauchy": X = rng.standard_cauchy(size=(T, n))     # heavy tails: the tail codes of the three-piece map
    elif kind == "t3": X = rng.standard_t(3, size=(T, n)) * 10.0
    elif kind == "levels": X = np.sort(rng.standard_t(3, size=n))[None, :] * 30.0 + X.cumsum(axis=0) * 0.05   # curves ordered by level
    elif kind == "spikes":                                            # a few per cent of the entries far outside the bulk, either side
        X = X.cumsum(axis=0); m = rng.random(size=X.shape) < 0.03; X[m] *= rng.choice([-1e6, 1e3, 1e6], size=int(m.sum()))
    elif kind == "tiny": X = X * 1e-310                               # denormals: the core's width underflows
    elif kind == "huge": X = X * 1e300 * rng.choice([1.0, 1e7])       # overflowing ranges
    elif kind == "clusters":                                          # tight clusters far apart: the sample's bracket sits in one of them
        X = X * 1e-6 + rng.choice([0.0, 1.0, 1e3], size=n, p=[0.8, 0.15, 0.05])[None, :]
    elif kind == "stride":                                            # a pattern in the curve index that the evenly spaced sample may miss or hit
        X = X.cumsum(axis=0); X[:, :: int(rng.choice([2, 5, 10, 11, 64]))] += rng.choice([50.0, 1e4])
    elif kind == "edge":                                              # many keys exactly at and a hair beyond the row's extremes
        X = X.cumsum(axis=0); hi = X.max(axis=1); lo = X.min(axis=1)
        for _ in range(40):
            r = int(rng.integers(0, T)); j = rng.integers(0, n, size=6)
            X[r, j[:2]] = hi[r]; X[r, j[2:4]] = np.nextafter(hi[r], np.inf); X[r, j[4]] = lo[r]; X[r, j[5]] = np.nextafter(lo[r], -np.inf)
    for _ in range(int(rng.integers(0, 6))):                          # image collisions / exact ties in single rows
        r = int(rng.integers(0, T)); a, b = rng.integers(0, n, size=2)
        X[r, a] = X[r, b] * (1 + rng.choice([0.0, 1e-16, 1e-15, 1e-13]))
    if rng.random() < 0.3: X[rng.integers(0, T, size=5), rng.integers(0, n, size=5)] = np.nan
    if rng.random() < 0.2: X[rng.integers(0, T), rng.integers(0, n)] = np.inf
    if rng.random() < 0.2: X[rng.integers(0, T), rng.integers(0, n)] = -np.inf
    if rng.random() < 0.3: X[:, rng.choice(n, size=3, replace=False)] *= rng.choice([1e3, 1e6])
    if rng.random() < 0.2: X[:, rng.integers(0, n)] = X[:, rng.integers(0, n)]          # a duplicated curve
    if rng.random() < 0.1:                                                              # 1 % duplicated curves
        src = rng.choice(n, size=n // 100, replace=False); X[:, (src + 1) % n] = X[:, src]
    if rng.random() < 0.15: X[rng.integers(0, T)] = 1.5
    if rng.random() < 0.15: X[rng.integers(0, T), ::2] = -0.0; 
    if rng.random() < 0.1: X[rng.integers(0, T)] = np.nan
    full = engine.mbd_counts(X, None, 2, algo="rank")
    tg = np.sort(rng.choice(n, size=160, replace=False))
    ref = engine.mbd_counts(X, tg, 2, algo="pairwise")
    if not (full[tg] == ref).all():
        bad += 1
        print(f"MISMATCH case {c}: n={n} T={T} kind={kind} first bad targets {tg[np.nonzero((full[tg] != ref).any(axis=1))[0][:5]]}", flush=True)
    if c % 10 == 9: print(f"{c + 1} cases, {bad} mismatches", flush=True)
print("FUZZ OK" if bad == 0 else f"FUZZ FAILED: {bad}")
sys.exit(1 if bad else 0)
