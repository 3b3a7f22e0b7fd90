"""Strict depth of short series (T = 4 ... 8; argv: "n,T" pairs): the state-class kernel against masks + matching (SD_STRICT_NOCLASS, cross-check library)."""
import os, sys, time, numpy as np
sys.path.insert(0, ".")
import torch
torch.cuda.init()
from statdepth_amd import engine, _native
if os.environ.get("SD_LIB"): _native.LIB_PATH = os.path.abspath(os.environ["SD_LIB"])     # experiments: another build of the library
PRODUCT = _native.load(); XCHECK = _native.open_library(_native.XCHECK_LIB_PATH)
CASES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(2000, 4), (10000, 4), (10000, 5), (100000, 4), (100000, 5), (30000, 5)]
for n, T in CASES:
    X = torch.from_numpy(np.random.default_rng(1).normal(size=(T, n))).cuda()
    res = []
    for force in (False, True):
        if force: os.environ.pop("SD_STRICT_NOCLASS", None); _native._LIB = PRODUCT     # state classes (the product's route for T <= 5)
        else: os.environ["SD_STRICT_NOCLASS"] = "1"; _native._LIB = XCHECK                # masks + matching
        a = engine.bd_strict_counts(X, None, 2, return_tensor=True) if "return_tensor" in engine.bd_strict_counts.__code__.co_varnames else engine.bd_strict_counts(X, None, 2)
        torch.cuda.synchronize(); t = time.perf_counter()
        a = engine.bd_strict_counts(X, None, 2); torch.cuda.synchronize()
        res.append(((time.perf_counter() - t) * 1e3, int(np.asarray(a).sum())))
    os.environ.pop("SD_STRICT_NOCLASS", None); _native._LIB = PRODUCT
    print(f"n={n} T={T}: masks+matching {res[0][0]:.2f} ms, classes {res[1][0]:.2f} ms, same={res[0][1]==res[1][1]}", flush=True)
