#!/usr/bin/env python3
"""Differential fuzz of the large-n rank route (n > 16 384; third generation: table partition, 8-byte records, 32-bit
ranking) against the pairwise kernel (independent code) on a sample of targets.  usage: fuzz_big.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from statdepth_amd import engine

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(cases):
    n = int(rng.choice([rng.integers(16385, 30000), rng.integers(30000, 120000), rng.integers(120000, 400000)], p=[0.5, 0.4, 0.1]))
    T = int(rng.integers(1, 7))
    kind = rng.choice(["normal", "walk", "ints", "round", "cauchy", "const", "lognormal", "tiny", "huge", "offset", "sorted"])
    X = rng.normal(size=(T, n))
    if kind == "walk": X = X.cumsum(axis=0)
    elif kind == "ints": X = rng.integers(-300, 400, size=(T, n)).astype(float)
    elif kind == "round": X = np.round(X * rng.choice([1, 10, 1000]), 0)
    elif kind == "cauchy": X = rng.standard_cauchy(size=(T, n))
    elif kind == "const": X[:] = rng.normal()
    elif kind == "lognormal": X = np.exp(X * 5)
    elif kind == "tiny": X = X * 1e-312
    elif kind == "huge": X = X * 1e307
    elif kind == "offset": X = 1e12 + X * rng.choice([1e-3, 1.0, 1e3])
    elif kind == "sorted": X = np.sort(X, axis=1)
    if rng.random() < 0.35: X[:, rng.choice(n, size=int(rng.choice([1, 3, 17, n // 50])), replace=False)] *= rng.choice([1e3, 1e6, 1e12])
    if rng.random() < 0.2: X[rng.random(X.shape) < 0.002] *= -1e8
    if rng.random() < 0.4:                       # near-collisions and sparse duplicates (image collisions)
        k = int(rng.choice([10, 300, 3000]))
        i = rng.choice(n, size=2 * k, replace=False)
        X[:, i[:k]] = X[:, i[k:]] * (1 + rng.choice([0.0, 2e-16, 1e-13]))
    if rng.random() < 0.25:
        w = int(rng.integers(2, 400))
        c0 = int(rng.integers(0, n - w))
        v = X[:, c0:c0 + 1].copy()
        X[:, c0:c0 + w] = np.where(rng.random((T, w)) < 0.5, v, v * (1 + rng.choice([0.0, 1e-15, 1e-12])))
    if rng.random() < 0.3:                       # rows with a large share on ONE value: the partition overflows a value bucket
        for r in rng.choice(T, size=int(rng.integers(1, T + 1)), replace=False):      # (chunked route inside the fall-back launch)
            X[r, rng.random(n) < rng.choice([0.1, 0.3, 0.7, 0.95])] = rng.normal()
    if rng.random() < 0.4: X[rng.random(X.shape) < rng.choice([0.0005, 0.02, 0.5])] = np.nan
    if rng.random() < 0.3: X[rng.random(X.shape) < 0.001] = np.inf
    if rng.random() < 0.3: X[rng.random(X.shape) < 0.001] = -np.inf
    tg = np.sort(rng.choice(n, size=300, replace=False))
    for J in (2, 3):
        a = engine.mbd_counts(X, None, J, algo="rank")[tg]
        b = engine.mbd_counts(X, tg, J, algo="pairwise")
        if not (a == b).all():
            bad += 1
            print(f"MISMATCH case {c}: n={n} T={T} kind={kind} J={J} first bad target {tg[np.nonzero((a != b).any(axis=1))[0][:5]]}", flush=True)
    if c % 10 == 9: print(f"{c + 1} cases, {bad} mismatches", flush=True)
print("FUZZ OK" if bad == 0 else f"FUZZ FAILED: {bad}")
sys.exit(1 if bad else 0)
