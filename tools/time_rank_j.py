#!/usr/bin/env python3
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from statdepth_amd import engine
n, T, J = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
X = np.random.default_rng(1234).normal(size=(T, n)).cumsum(axis=0)
Xd = engine.to_device_matrix(X)
for _ in range(3):
    engine.mbd_counts(Xd, None, J, algo="rank", return_tensor=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    engine.mbd_counts(Xd, None, J, algo="rank", return_tensor=True)
e1.record(); torch.cuda.synchronize()
print(f"n={n} T={T} J={J}: {e0.elapsed_time(e1)/20:.4f} ms per call")
