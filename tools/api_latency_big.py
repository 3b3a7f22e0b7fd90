#!/usr/bin/env python3
"""End-to-end FunctionalDepth latency (DataFrame in -> Series out) at config-2 size, with a breakdown."""
import sys, time, os
import numpy as np, pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from statdepth_amd import FunctionalDepth, engine
rng = np.random.default_rng(0)
T, n = 1000, 10000
X = rng.normal(size=(T, n)).cumsum(axis=0)
df = pd.DataFrame(X)
def tm(f, reps=5):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3
print(f"FunctionalDepth([df], relax=True)      : {tm(lambda: FunctionalDepth([df], relax=True)):8.3f} ms")
print(f"df.to_numpy()                           : {tm(lambda: df.to_numpy()):8.3f} ms")
A = df.to_numpy()
print(f"engine.to_device_matrix(ndarray)  (H2D) : {tm(lambda: engine.to_device_matrix(A)):8.3f} ms")
Xd = engine.to_device_matrix(A)
print(f"engine.mbd_counts(resident) -> numpy    : {tm(lambda: engine.mbd_counts(Xd, None, 2)):8.3f} ms")
print(f"engine.mbd_counts(resident) -> tensor   : {tm(lambda: engine.mbd_counts(Xd, None, 2, return_tensor=True)):8.3f} ms")
pin = torch.from_numpy(A).pin_memory()
print(f"pinned H2D                              : {tm(lambda: pin.to('cuda', non_blocking=True)):8.3f} ms")
