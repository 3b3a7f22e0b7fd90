#!/usr/bin/env python3
"""End-to-end latency of FunctionalDepth (DataFrame in -> Series out) at small sizes, per algorithm."""
import sys, time, os
import numpy as np, pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from statdepth_amd import FunctionalDepth, engine
rng = np.random.default_rng(0)
for (T, n) in [(14, 160), (100, 50), (100, 200), (100, 1000), (1000, 1000), (1000, 3000)]:
    df = pd.DataFrame(rng.normal(size=(T, n)).cumsum(axis=0))
    Xd = engine.to_device_matrix(df.to_numpy())
    row = [f"{T}x{n}"]
    for algo in ("pairwise", "rank", "auto"):
        FunctionalDepth([df], relax=True, algo=algo)
        t = time.perf_counter()
        for _ in range(20):
            FunctionalDepth([df], relax=True, algo=algo)
        api = (time.perf_counter() - t) / 20
        engine.mbd_counts(Xd, None, 2, algo=algo, return_tensor=True); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(50):
            engine.mbd_counts(Xd, None, 2, algo=algo, return_tensor=True)
        torch.cuda.synchronize()
        dev = (time.perf_counter() - t) / 50
        row.append(f"{algo}: api {api*1e3:.3f} ms, resident {dev*1e3:.3f} ms")
    print(" | ".join(row))
