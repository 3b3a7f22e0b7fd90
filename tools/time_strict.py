#!/usr/bin/env python3
"""Device time of sd_bd_strict_counts (J=2) on n curves x T timepoints, all targets."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from statdepth_amd import engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
X = np.random.default_rng(3).normal(size=(T, n)).cumsum(axis=0)
engine.bd_strict_counts(X, None, 2)
torch.cuda.synchronize()
import time
t = time.perf_counter()
for _ in range(3):
    engine.bd_strict_counts(X, None, 2)
torch.cuda.synchronize()
print(f"n={n} T={T}: {(time.perf_counter() - t) / 3 * 1e3:.2f} ms per call")
