#!/usr/bin/env python3
"""Config 4 (i): 5 000 curves x 500 timepoints x 8 features, S sampled 9-point simplices per target (relax=True), checked on a few
targets against the oracle.  argv: S (default 4096)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle
from statdepth_amd import _native
from statdepth_amd._native import check
lib = _native.load()
dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n4, T4, d4 = 5000, 500, 8
g = torch.Generator(device=dev).manual_seed(1236)
C4 = torch.randn(n4, T4, d4, dtype=torch.float64, device=dev, generator=g).cumsum(1)
o4 = torch.empty(n4, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
USE_WS = os.environ.get('SD_NO_WS') is None
wss = int(lib.sd_simplex_sampled_workspace_bytes(n4, T4, d4, S)) if USE_WS else 0
ws = torch.empty(max(wss, 8), dtype=torch.uint8, device=dev)
for rep in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    check(lib.sd_multi_simplex_sampled(C4.data_ptr(), n4, T4, d4, 0, n4, 1, 1e-7, S, 1236, o4.data_ptr(), ws.data_ptr(), wss, st))
    torch.cuda.synchronize(); dt = time.perf_counter() - t
tg = np.array([0, 2500, 4999])
Sc = min(S, 64)
check(lib.sd_multi_simplex_sampled(C4.data_ptr(), n4, T4, d4, 0, n4, 1, 1e-7, Sc, 1236, o4.data_ptr(), ws.data_ptr(), wss, st))
oracle.build()
want = oracle.simplex_sampled(C4.cpu().numpy(), tg, relax=True, samples=Sc, seed=1236)
print(f"config 4 (i) S={S} workspace {wss >> 20} MiB: {dt * 1e3:.1f} ms, {n4 * S * T4 / dt:.3e} tests/s; oracle check at S={Sc}: {(o4.cpu().numpy()[tg] == want).all()}", flush=True)
