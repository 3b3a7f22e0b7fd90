#!/usr/bin/env python3
"""Per-workgroup durations of rank_bucket32_kernel (timing experiments, GPU box only): runs one config-2-shaped call on a
library built with -DR32_STAMPS (tools/build_variant.sh st -DR32_STAMPS, SRC=mbd_rank_bucket32), which leaves
(cycles, start, end in 100 MHz ticks, set-aside keys | rows << 16) per workgroup behind the gate word of the workspace."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from statdepth_amd import engine, _native
_native.LIB_PATH = os.path.abspath(os.environ.get("SD_LIB", "statdepth_amd/lib/libsd_st.so"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
X = np.random.default_rng(1234).normal(size=(T, n)).cumsum(axis=0)
Xd = engine.to_device_matrix(X)
for _ in range(3):
    engine.mbd_counts(Xd, None, 2, algo="rank", return_tensor=True)
torch.cuda.synchronize()
ws = list(engine._ws_cache.values())[0]
cus = torch.cuda.get_device_properties(0).multi_processor_count
G = min(T, 2 * cus)
off = 2 * cus * n * 4 + (T + 63) // 64 * 64 + 64 + G * 129 * 4
d = ws.view(torch.uint8)[off:off + G * 16].cpu().numpy().view(np.uint32).reshape(G, 4).astype(np.int64)
t0 = d[:, 1].min()
dur = (d[:, 2] - d[:, 1]) / 100.0
print(f"G={G} cycles min/med/max {d[:,0].min()} {int(np.median(d[:,0]))} {d[:,0].max()}")
print(f"start us min/med/max {((d[:,1]-t0)/100).min():.2f} {np.median((d[:,1]-t0)/100):.2f} {((d[:,1]-t0)/100).max():.2f}")
print(f"end   us min/med/max {((d[:,2]-t0)/100).min():.2f} {np.median((d[:,2]-t0)/100):.2f} {((d[:,2]-t0)/100).max():.2f}")
print(f"dur   us min/med/max {dur.min():.2f} {np.median(dur):.2f} {dur.max():.2f}; percentiles 10/50/90/99: {np.percentile(dur,[10,50,90,99]).round(2)}")
print("clock GHz (cycles / duration):", np.median(d[:, 0] / (dur * 1e3)).round(3))
lst = d[:, 3] & 0xFFFF
rows = d[:, 3] >> 16
print("workgroups with set-aside keys:", int((lst > 0).sum()), "their median dur", np.median(dur[lst > 0]) if (lst > 0).any() else None)
for k in sorted(set(rows)):
    print(f"  rows={k}: {int((rows==k).sum())} workgroups, dur med {np.median(dur[rows==k]):.2f} max {dur[rows==k].max():.2f}")
o = np.argsort(dur)[-8:]
print("slowest:", [(int(i), round(float(dur[i]), 2), int(lst[i])) for i in o])
slow = np.nonzero(dur > np.percentile(dur, 80))[0]
print("slow workgroups (> 80th pct):", slow.tolist())
print("dur by index block of 32:", [round(float(np.median(dur[i:i+32])),1) for i in range(0, G, 32)])
print("starts of slow (us):", np.round((d[slow,1]-t0)/100.0,2).tolist()[:40])
