#!/usr/bin/env python3
"""Fuzz of the entry points added in round 2 against the CPU oracle (GPU box): componentwise band containment (K6),
external / blocked point-cloud targets (simplex, L1), strict band depth by complement matching, two-limb totals.  usage: fuzz_round2.py [cases] [seed]"""
import math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from statdepth_amd import engine
oracle.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(cases):
    # ---- K6: state-class pair counting against the literal enumeration ----
    d = int(rng.integers(1, 9)); n = int(rng.integers(3, 40)); T = int(rng.integers(1, 9))
    P = rng.normal(size=(n, T, d)).cumsum(axis=1)
    if rng.random() < 0.6: P = np.round(P, int(rng.integers(0, 2)))          # ties
    if rng.random() < 0.3: P[rng.integers(0, n)] = P[rng.integers(0, n)]
    if rng.random() < 0.2: P[:, :, rng.integers(0, d)] = 0.5                   # a constant feature
    if not (engine.multi_band_counts(P) == oracle.multi_band_enum(P, None, 2, True)[:, 0]).all():
        bad += 1; print(f"K6 MISMATCH case {c}: n={n} T={T} d={d}", flush=True)
    # ---- point clouds: external targets and explicit blocks ----
    d = int(rng.integers(1, 6)); n = int(rng.integers(d + 2, d + 12)); m = int(rng.integers(1, 9))
    F = rng.normal(size=(n, d)); G = rng.normal(size=(m, d)) * 0.7
    if rng.random() < 0.3: F = np.round(F, 0); G = np.round(G, 0)
    cnt = engine.pointcloud_simplex_external_counts(F, G); l1 = engine.l1_external_depth(F, G)
    for q in range(m):
        Fg = np.concatenate([F, G[q:q + 1]])
        w1 = oracle.l1_depth(Fg, [n])[0]
        ok = cnt[q] == oracle.pointcloud_simplex_counts(Fg, [n])[0] and (np.isnan(w1) == np.isnan(l1[q])) and \
            (np.isnan(w1) or abs(w1 - l1[q]) <= 1e-12)
        if not ok:
            bad += 1; print(f"EXTERNAL MISMATCH case {c}: n={n} d={d} q={q}", flush=True)
    blocks = [rng.choice(n, size=int(rng.integers(d + 2, n + 1)), replace=False) for _ in range(6)]
    M = np.full((6, max(len(b) for b in blocks)), -1, dtype=np.int32)
    for i, b in enumerate(blocks): M[i, :len(b)] = b
    cnt = engine.pointcloud_simplex_subset_counts(F, M)
    for i, b in enumerate(blocks):
        if cnt[i] != oracle.pointcloud_simplex_counts(F[b], [len(b) - 1])[0]:
            bad += 1; print(f"SUBSET MISMATCH case {c}: n={n} d={d} block {i}", flush=True)
    # ---- strict band depth: complement matching with every kind of curve mixed in ----
    T = int(rng.choice([1, 2, 3, 4, 17, 32, 33, 64, 95, 130, 257, 1025, 1100])); n = int(rng.integers(4, 260))
    if rng.random() < 0.5:
        X = np.sort(rng.normal(size=n))[None, :] * rng.choice([0.5, 3.0]) + rng.normal(size=(T, n)) * rng.choice([0.05, 0.5])
    else:
        X = rng.normal(size=(T, n)).cumsum(axis=0)
    for _ in range(int(rng.integers(0, 4))):                  # groups sharing a crossing pattern with some curve, both sides
        base = int(rng.integers(0, n)); sign = np.where(rng.random(T) < 0.5, 1.0, -1.0)
        for i in rng.choice(n, size=min(n, int(rng.integers(1, 6))), replace=False):
            if i != base: X[:, i] = X[:, base] + sign * rng.choice([-1.0, 1.0]) * rng.uniform(0.01, 0.5)
    if rng.random() < 0.3: X[rng.integers(0, T)] = 0.25       # a constant row
    if rng.random() < 0.2: X[0] = X[0, 0]
    if rng.random() < 0.3:                                    # sparse ties
        for _ in range(int(rng.integers(1, 5))):
            i, j, t = rng.integers(0, n), rng.integers(0, n), rng.integers(0, T); X[t, i] = X[t, j]
    if rng.random() < 0.15: X[:, rng.integers(0, n)] = X[:, rng.integers(0, n)]
    if rng.random() < 0.15: X[rng.integers(0, T), rng.integers(0, n)] = np.nan
    if rng.random() < 0.1: X = np.round(X, 1)                 # ties everywhere
    tg = rng.choice(n, size=min(n, 12), replace=False)
    if not (engine.bd_strict_counts(X, tg, 2)[:, 0] == oracle.bd_strict_counts(X, tg)).all():
        bad += 1; print(f"STRICT MISMATCH case {c}: n={n} T={T}", flush=True)
    # ---- two-limb totals ----
    if c % 10 == 0:
        n = int(rng.integers(1500, 3200)); J = 6; T = int(rng.integers(12, 40))
        X = np.round(rng.normal(size=(T, n)).cumsum(axis=0), 1)
        tg = rng.choice(n, size=3, replace=False)
        got = engine.mbd_counts_wide(X, tg, J)
        for qi, q in enumerate(tg):
            A = (X > X[:, q:q + 1]).sum(axis=1); B = (X < X[:, q:q + 1]).sum(axis=1)
            want = [sum(math.comb(n - 1, j) - math.comb(int(a), j) - math.comb(int(b), j) for a, b in zip(A, B)) for j in range(2, J + 1)]
            if [int(v) for v in got[qi]] != want:
                bad += 1; print(f"WIDE MISMATCH case {c}: n={n} T={T}", flush=True)
    if c % 20 == 19: print(f"{c + 1} cases, {bad} mismatches", flush=True)
print("FUZZ OK" if bad == 0 else f"FUZZ FAILED: {bad}")
sys.exit(1 if bad else 0)
