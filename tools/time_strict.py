#!/usr/bin/env python3
"""Device time of sd_bd_strict_counts (J=2) on n curves x T timepoints, all targets.
usage: time_strict.py [n] [T] [kind]
kind: walks (default) | banded | rounded (ties everywhere) | start0 (walks from a common start) | fewties (walks, 1 % of
the curves touch another curve once)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from statdepth_amd import engine, _native
if os.environ.get("SD_LIB"):                     # experiments: another build of the library (this tool only)
    _native.LIB_PATH = os.path.abspath(os.environ["SD_LIB"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
kind = sys.argv[3] if len(sys.argv) > 3 else "walks"
rng = np.random.default_rng(3)
if kind in ("walks", "start0", "fewties"):
    X = rng.normal(size=(T, n)).cumsum(axis=0)
    if kind == "start0":
        X -= X[0]
    if kind == "fewties":
        for c in rng.choice(n, max(1, n // 100), replace=False):
            t = rng.integers(T)
            X[t, c] = X[t, (c + 1) % n]
elif kind == "banded":
    X = np.sort(rng.normal(size=n))[None, :] * 3.0 + rng.normal(size=(T, n)) * 0.3
else:
    X = np.round(np.sort(rng.normal(size=n))[None, :] * 3.0 + rng.normal(size=(T, n)) * 0.3, 1)
Xd = torch.from_numpy(X).cuda()
out = engine.bd_strict_counts(Xd, None, 2)
torch.cuda.synchronize()
reps = 3
t = time.perf_counter()
for _ in range(reps):
    engine.bd_strict_counts(Xd, None, 2)
torch.cuda.synchronize()
print(f"strict n={n} T={T} {kind}: {(time.perf_counter() - t) / reps * 1e3:.2f} ms per call, contained pairs {int(out.sum())}")
