#!/usr/bin/env python3
"""World-size-1 RCCL run of the time-sharded path with a per-section device timeline (GPU box only)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from statdepth_amd import distributed as D, engine
n, T = 10000, 1000
X = torch.from_numpy(np.random.default_rng(1).normal(size=(T, n)).cumsum(axis=0)).cuda()
for K in (1, 2, 4):
    for _ in range(3): D.sharded_mbd_counts_time(X, J=2, sizes=[n], chunks=K, _force_exchange=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): D.sharded_mbd_counts_time(X, J=2, sizes=[n], chunks=K, _force_exchange=True)
    torch.cuda.synchronize(); print(f"K={K}: {(time.perf_counter()-t0)/20*1e3:.3f} ms per call", flush=True)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5): D.sharded_mbd_counts_time(X, J=2, sizes=[n], chunks=2, _force_exchange=True)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
dist.destroy_process_group()
