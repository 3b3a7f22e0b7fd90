#!/usr/bin/env python3
"""Development check of the bucket rank kernel (the product library's default) on the GPU box: equality with the
packed-sort path (SD_RANK_IMPL=3 in the cross-check library) on whole matrices and with the CPU oracle on a target
sample; device time per call."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle
torch.cuda.init()
from statdepth_amd import engine, _native
PRODUCT = _native.load()
XCHECK = _native.open_library(_native.XCHECK_LIB_PATH)

oracle.build()


def run(X, J, impl, reps=0):
    if impl == 4:
        os.environ.pop("SD_RANK_IMPL", None)
        _native._LIB = PRODUCT
    else:
        os.environ["SD_RANK_IMPL"] = str(impl)
        _native._LIB = XCHECK
    Xd = engine.to_device_matrix(X)
    out = engine.mbd_counts(Xd, None, J, algo="rank")
    ms = None
    if reps:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            engine.mbd_counts(Xd, None, J, algo="rank", return_tensor=True)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
    return out, ms


def case(name, X, J=2, reps=0, oracle_targets=64):
    a, ms4 = run(X, J, 4, reps)
    b, ms3 = run(X, J, 3, reps)
    ok = (a == b).all()
    tg = np.linspace(0, X.shape[1] - 1, oracle_targets).astype(np.int64)
    if not ok:
        bad = np.nonzero((a != b).any(axis=1))[0]
        print("   differing targets:", bad[:20], "of", len(bad))
        tg = np.unique(np.concatenate([tg, bad[:64]]))
        w = oracle.mbd_counts(X, bad[:8], J)
        print("   impl4", a[bad[:8]].ravel(), "\n   impl3", b[bad[:8]].ravel(), "\n   oracle", w.ravel())
    want = oracle.mbd_counts(X, tg, J)
    ok2 = (a[tg] == want).all()
    print(f"{name:40s} T={X.shape[0]:5d} n={X.shape[1]:6d} J={J} vs impl3 {'OK' if ok else 'DIFF'} vs oracle "
          f"{'OK' if ok2 else 'DIFF'}" + (f"  impl4 {ms4:.4f} ms  impl3 {ms3:.4f} ms" if reps else ""), flush=True)
    return ok and ok2


def main():
    rng = np.random.default_rng(5)
    good = True
    for n in (2, 3, 17, 64, 65, 300, 1023, 1024, 1025, 2049, 4096, 5000, 7777, 8192, 8193, 9000, 10000, 10240, 10241, 12000, 13313,
              15000, 16384):
        X = rng.normal(size=(37, n)).cumsum(axis=0)
        good &= case("walk", X, oracle_targets=min(64, n))
        if n < 9000:
            Xt = np.round(X, 0); Xt[3, ::5] = np.nan; Xt[4, 1] = np.inf
            good &= case("ints+nan", Xt, J=3 if n > 3 else 2, oracle_targets=min(64, n))
    X = rng.normal(size=(64, 10000)).cumsum(axis=0)
    good &= case("walk J=3", X, J=3)
    good &= case("ties (0.1)", np.round(X, 1))
    good &= case("heavy ties (integers)", np.round(X, 0))
    Y = X.copy(); Y[3, 17] = np.nan; Y[5, :40] = np.nan; Y[9, 100] = np.inf; Y[11, 7] = -np.inf; Y[12, 5:9] = np.inf; Y[12, 20:23] = -np.inf; Y[12, 30] = np.nan; Y[13, :] = np.nan; Y[14, 1:] = np.nan; Y[15, 2:] = np.inf
    good &= case("NaN / inf rows", Y)
    Y = X.copy(); Y[4, :] = 1.25; Y[6, :] = 0.0
    good &= case("constant rows", Y)
    Y = X.copy(); Y[:, 0] = 1e300; Y[:, 1] = -1e300
    good &= case("huge outliers (range ~ 2e300)", Y)
    Y = X.copy(); Y[:, 0] = 1.7e308; Y[:, 1] = -1.7e308
    good &= case("range overflows", Y)
    Y = X * 1e-310
    good &= case("denormals", Y)
    Y = X.copy(); Y[:, ::2] = Y[:, 1::2]
    good &= case("every curve duplicated", Y)
    Y = np.exp(X)                      # skewed
    good &= case("log-normal (skewed)", Y)
    Y = rng.standard_cauchy(size=(64, 10000))
    good &= case("cauchy (heavy tails)", Y)
    X2 = rng.normal(size=(1000, 10000)).cumsum(axis=0)
    good &= case("config 2", X2, reps=20)
    good &= case("config 2 ties", np.round(X2, 1), reps=20)
    X3 = rng.normal(size=(500, 16384)).cumsum(axis=0)
    good &= case("n=16384", X3, reps=10)
    print("ALL OK" if good else "FAILURES")
    return 0 if good else 1


if __name__ == "__main__":
    sys.exit(main())
