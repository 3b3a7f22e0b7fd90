#!/usr/bin/env python3
"""Medium route (two column blocks per workgroup) against the large-n route at the same n (GPU box; cross-check library)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from statdepth_amd import _native, engine
_native._LIB = _native.open_library(_native.XCHECK_LIB_PATH)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 500
X = torch.from_numpy(np.random.default_rng(1).normal(size=(T, n)).cumsum(axis=0)).cuda()
for name, env in (("medium", None), ("large-n", "1")):
    if env:
        os.environ["SD_BIG_NOMEDIUM"] = env
    else:
        os.environ.pop("SD_BIG_NOMEDIUM", None)
    engine.mbd_counts(X, None, J=2)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        engine.mbd_counts(X, None, J=2)
    torch.cuda.synchronize()
    print(f"n={n} T={T} {name}: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms per call (host call included)")
