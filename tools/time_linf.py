#!/usr/bin/env python3
"""L-infinity (box) depth with relax=False of n points in R^3, every point a target: the grid of cells (bd_strict_grid.hip) against
the state-class kernel (SD_STRICT_NOGRID, cross-check library).  argv: n values (default 100000 300000 1000000); SD_KIND=ties
rounds the coordinates to one decimal."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from statdepth_amd import engine, _native
PRODUCT = _native.load()
XCHECK = _native.open_library(_native.XCHECK_LIB_PATH)
D = int(os.environ.get("SD_D", "3"))
for n in [int(a) for a in sys.argv[1:]] or [100000, 300000, 1000000]:
    P = np.random.default_rng(1237).normal(size=(n, D))
    if os.environ.get("SD_KIND") == "ties":
        P = np.round(P, 1)
    X = torch.from_numpy(np.ascontiguousarray(P.T)).cuda()
    res = {}
    for name in ("grid", "classes"):
        if name == "classes":
            if n > 400000 and not os.environ.get("SD_BOTH"):
                continue
            os.environ["SD_STRICT_NOGRID"] = "1"; _native._LIB = XCHECK
        else:
            os.environ.pop("SD_STRICT_NOGRID", None); _native._LIB = PRODUCT
        engine.bd_strict_counts(X, None, 2)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(3):
            a = engine.bd_strict_counts(X, None, 2)
        torch.cuda.synchronize()
        res[name] = ((time.perf_counter() - t) / 3 * 1e3, int(np.asarray(a).sum()))
    os.environ.pop("SD_STRICT_NOGRID", None); _native._LIB = PRODUCT
    print(f"n={n}: " + ", ".join(f"{k} {v[0]:.2f} ms" for k, v in res.items()) + (f", same={res['grid'][1] == res['classes'][1]}" if len(res) == 2 else ""), flush=True)
