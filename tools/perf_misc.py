import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from statdepth_amd import engine
def tm(f, reps=2):
    f(); torch.cuda.synchronize()
    t=time.time()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.time()-t)/reps
rng=np.random.default_rng(0)
# pointcloud simplex exhaustive n=200 d=2
P=rng.normal(size=(200,2)); t=tm(lambda: engine.pointcloud_simplex_counts(P)); import math
print("pc simplex n=200 d=2: %.3f s, %.3g tests/s"%(t, 200*math.comb(199,3)/t))
P=rng.normal(size=(60,3)); t=tm(lambda: engine.pointcloud_simplex_counts(P))
print("pc simplex n=60 d=3: %.3f s, %.3g tests/s"%(t, 60*math.comb(59,4)/t))
# sampled: config-5-like n=1e5 d=3 R=256
P=rng.normal(size=(100000,3)); t=tm(lambda: engine.pointcloud_simplex_counts(P, samples=256, seed=1), 1)
print("pc sampled n=1e5 d=3 R=256: %.3f s, %.3g tests/s"%(t, 1e5*256/t))
# multi sampled config-4-like: n=500 curves, T=50, d=8, S=256
C=rng.normal(size=(500,50,8)).cumsum(axis=1); t=tm(lambda: engine.multi_simplex_counts(C, samples=256, seed=1), 1)
print("multi sampled n=500 T=50 d=8 S=256: %.3f s, %.3g tests/s"%(t, 500*256*50/t))
# l1
P=rng.normal(size=(100000,3)); t=tm(lambda: engine.l1_depth(P), 1)
print("l1 n=1e5 d=3: %.3f s, %.3g pairs/s"%(t, 1e10/t))
# strict
X=np.round(np.sort(rng.normal(size=2000))[None,:]*3+rng.normal(size=(1000,2000))*0.3,1); t=tm(lambda: engine.bd_strict_counts(X), 1)
print("strict n=2000 T=1000: %.3f s, %.3g pair-tests/s"%(t, 2000*math.comb(1999,2)/t))
