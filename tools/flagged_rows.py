#!/usr/bin/env python3
"""Which rows of a config-2-shaped call does rank_bucket32_kernel leave to the second launch?  (timing experiments, GPU box
only: reads the row flags behind the partial blocks of the engine's cached workspace)  usage: flagged_rows.py n T; data variants
through the environment as in tools/_rankdata.py."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from statdepth_amd import engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
from _rankdata import rank_data
X = rank_data(n, T)
Xd = engine.to_device_matrix(X)
engine.mbd_counts(Xd, None, 2, algo="rank", return_tensor=True)
torch.cuda.synchronize()
ws = list(engine._ws_cache.values())[-1]
cus = torch.cuda.get_device_properties(0).multi_processor_count
imgb = 2 * cus * ((n + 3) // 4 * 4) * 4
flags = ws.view(torch.uint8)[imgb:imgb + T].cpu().numpy()
idx = np.nonzero(flags)[0]
print(f"n={n} T={T}: {len(idx)} rows flagged; first {idx[:20].tolist()} last {idx[-5:].tolist()}")
for r in list(idx[:3]) + [int(np.nonzero(flags == 0)[0][0]) if (flags == 0).any() else 0]:
    v, c = np.unique(X[r], return_counts=True)
    print(f"  row {r} flagged={int(flags[r])}: {len(v)} distinct values, largest group {c.max()}, groups >= 16: {(c >= 16).sum()}, >= 64: {(c >= 64).sum()}, "
          f"range {X[r].min():.1f} .. {X[r].max():.1f}")
