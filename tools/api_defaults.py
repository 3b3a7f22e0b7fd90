import sys, time, os
import numpy as np, pandas as pd
sys.path.insert(0, os.getcwd())
import torch
from statdepth_amd import FunctionalDepth, PointcloudDepth
rng = np.random.default_rng(0)
for (T, n) in [(100, 500), (500, 3000), (1000, 5000)]:
    X = np.sort(rng.normal(size=n))[None, :] * 3 + rng.normal(size=(T, n)).cumsum(axis=0) * 0.05
    df = pd.DataFrame(X)
    for relax in (False, True):
        FunctionalDepth([df], relax=relax)
        torch.cuda.synchronize(); t = time.perf_counter()
        d = FunctionalDepth([df], relax=relax)
        torch.cuda.synchronize()
        print(f"FunctionalDepth T={T} n={n} relax={relax}: {(time.perf_counter()-t)*1e3:.2f} ms  deepest={d.deepest(1).index[0]}", flush=True)
P = pd.DataFrame(rng.normal(size=(2000, 3)))
for c in ("l1",):
    PointcloudDepth(P, containment=c); torch.cuda.synchronize(); t = time.perf_counter()
    PointcloudDepth(P, containment=c); torch.cuda.synchronize()
    print(f"PointcloudDepth n=2000 d=3 {c}: {(time.perf_counter()-t)*1e3:.2f} ms")
P = pd.DataFrame(rng.normal(size=(120, 2)))
PointcloudDepth(P, containment="simplex"); torch.cuda.synchronize(); t = time.perf_counter()
PointcloudDepth(P, containment="simplex"); torch.cuda.synchronize()
print(f"PointcloudDepth n=120 d=2 simplex (exhaustive): {(time.perf_counter()-t)*1e3:.2f} ms")
