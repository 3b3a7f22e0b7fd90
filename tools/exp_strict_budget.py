#!/usr/bin/env python3
"""Strict depth (J = 2) at n x T under workspace budgets: do batches of targets whose masks fit the 256 MB Infinity Cache
(mask kernel writes, digest kernel reads them back) beat the 2 GiB batches?  usage: exp_strict_budget.py [n] [T] [MiB ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from statdepth_amd import engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
budgets = [int(a) for a in sys.argv[3:]] or [0, 1024, 512, 256, 128, 64]
X = np.random.default_rng(1234).normal(size=(T, n)).cumsum(axis=0)
Xd = engine.to_device_matrix(X)
ref = None
for mb in budgets:
    kw = {} if mb == 0 else {"workspace_budget": mb << 20}
    engine.release_workspace()
    got = engine.bd_strict_counts(Xd, None, 2, **kw)
    if ref is None: ref = got
    assert (got == ref).all(), mb
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        engine.bd_strict_counts(Xd, None, 2, **kw)
    e1.record()
    torch.cuda.synchronize()
    print(f"n={n} T={T} budget {mb or 'default'} MiB: {e0.elapsed_time(e1) / 3:.3f} ms per call (incl. D2H of the counts)", flush=True)
