import numpy as np, sys
sys.path.insert(0,'/root/repo')
import oracle
from statdepth_amd import engine
rng=np.random.default_rng(1)
for (T,n) in [(1,16384+16384+5),(1,33000),(2,33000),(1,50000)]:
    X=rng.normal(size=(T,n))
    got=engine.mbd_counts(X,None,2,algo='rank')[:,0]
    want=oracle.mbd_counts(X,None,2)[:,0]
    bad=np.nonzero(got!=want)[0]
    print(T,n,'bad',len(bad), bad[:10], (got-want)[bad[:10]])
    if len(bad):
        ab=oracle.above_below(X, bad[:3])
        print(' chunks of bad', np.unique(bad//16384, return_counts=True))
