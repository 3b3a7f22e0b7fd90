#!/usr/bin/env python3
"""Device time of sd_mbd_counts (rank path) on config-2-shaped data under env-selected variants.
usage: time_rank.py [n] [T] [reps]; prints ms per call for each SD_RB_DBG level given in SD_LEVELS (default 0)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from statdepth_amd import engine, _native
if os.environ.get("SD_LIB"):                     # experiments: another build of the library (this tool only; the cross-check
    _native.LIB_PATH = os.path.abspath(os.environ["SD_LIB"])      # build too, which the package itself refuses to load)
    _native._LIB = _native.open_library(_native.LIB_PATH)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
from _rankdata import rank_data
X = rank_data(n, T)
Xd = engine.to_device_matrix(X)
ROT = int(os.environ.get("SD_ROTATE", "1"))      # > 1: that many distinct matrices in rotation (every call streams from HBM)
Xs = [Xd] + [engine.to_device_matrix(X + float(k)) for k in range(1, ROT)]
if os.environ.get("SD_RANK_IMPL"):
    print("SD_RANK_IMPL =", os.environ["SD_RANK_IMPL"])
for lvl in os.environ.get("SD_LEVELS", "0").split(","):
    if lvl != "0":
        os.environ["SD_RB_DBG"] = lvl
    else:
        os.environ.pop("SD_RB_DBG", None)
    for _ in range(3):
        engine.mbd_counts(Xd, None, 2, algo="rank", return_tensor=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        engine.mbd_counts(Xs[i % ROT], None, 2, algo="rank", return_tensor=True)
    e1.record()
    torch.cuda.synchronize()
    print(f"n={n} T={T} dbg={lvl}: {e0.elapsed_time(e1) / reps:.4f} ms per call", flush=True)
