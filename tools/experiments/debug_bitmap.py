#!/usr/bin/env python3
"""Development check of rank_bitmap_kernel: per-curve differences against the oracle and how many rows were set aside."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle
from statdepth_amd import engine, _native
from statdepth_amd._native import check
oracle.build()
lib = _native.require_device()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2049
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4
kind = sys.argv[3] if len(sys.argv) > 3 else "normal"
rng = np.random.default_rng(7)
X = rng.normal(size=(T, n))
if kind == "walk":
    X = X.cumsum(axis=0)
Xd = torch.from_numpy(X).cuda()
out = torch.empty((n, 1), dtype=torch.int64, device="cuda")
wsb = lib.sd_mbd_workspace_bytes(T, n, n, 1, n, 2, 2)
ws = torch.zeros(int(wsb), dtype=torch.uint8, device="cuda")
check(lib.sd_mbd_counts(Xd.data_ptr(), T, n, n, 1, 0, n, 2, 2, out.data_ptr(), ws.data_ptr(), wsb, 0))
torch.cuda.synchronize()
got = out.cpu().numpy()[:, 0]
want = oracle.mbd_counts(X, None, 2)[:, 0]
bad = np.nonzero(got != want)[0]
print(f"n={n} T={T} {kind}: {len(bad)} curves differ", bad[:20], (got - want)[bad[:20]])
# rank workspace: [partial 512 * n * 8 aligned][rowsel][wgdefer]
off = 0
pb = (512 * n * 8 + 255) // 256 * 256
w = ws.cpu().numpy()
rowsel = w[off + pb: off + pb + T]
wg = w[off + pb + (T + 255) // 256 * 256:][:4 * min(T, 256)].view(np.uint32)
print("rows set aside:", int(rowsel.sum()), "of", T, "| per-workgroup counts (first 16):", wg[:16])
