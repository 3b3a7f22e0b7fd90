#!/usr/bin/env python3
"""Per-workgroup durations and phase cycles of rank_bucket32_kernel (timing experiments, GPU box only): runs one
config-2-shaped call on a library built with -DR32_STAMPS (SRC=mbd_rank_bucket32 tools/build_variant.sh st -DR32_STAMPS), which
leaves (cycles, start, end in 100 MHz ticks, set-aside keys | rows << 16, ten phase stamps) for the oldest and the youngest
wave of every workgroup behind the slice flags of the workspace."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from statdepth_amd import engine, _native
_native.LIB_PATH = os.path.abspath(os.environ.get("SD_LIB", "statdepth_amd/lib/libsd_st.so"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
from _rankdata import rank_data
X = rank_data(n, T)
Xd = engine.to_device_matrix(X)
for _ in range(3):
    engine.mbd_counts(Xd, None, 2, algo="rank", return_tensor=True)
torch.cuda.synchronize()
ws = list(engine._ws_cache.values())[-1]
cus = torch.cuda.get_device_properties(0).multi_processor_count
G = min(T, 2 * cus)
off = 2 * cus * ((n + 3) // 4 * 4) * 4 + (T + 63) // 64 * 64 + 64 + G * 129 * 4
d2 = ws.view(torch.uint8)[off:off + G * 128].cpu().numpy().view(np.uint32).reshape(G, 2, 16).astype(np.int64)
d = d2[:, 0, :4]
imgb = 2 * cus * ((n + 3) // 4 * 4) * 4
flags = ws.view(torch.uint8)[imgb:imgb + T].cpu().numpy()
print('rows flagged for the second launch:', int(flags.sum()), 'of', T)
t0 = d[:, 1].min()
dur = (d[:, 2] - d[:, 1]) / 100.0
print(f"G={G} cycles min/med/max {d[:,0].min()} {int(np.median(d[:,0]))} {d[:,0].max()}")
print(f"start us min/med/max {((d[:,1]-t0)/100).min():.2f} {np.median((d[:,1]-t0)/100):.2f} {((d[:,1]-t0)/100).max():.2f}")
print(f"end   us min/med/max {((d[:,2]-t0)/100).min():.2f} {np.median((d[:,2]-t0)/100):.2f} {((d[:,2]-t0)/100).max():.2f}")
print(f"dur   us min/med/max {dur.min():.2f} {np.median(dur):.2f} {dur.max():.2f}; percentiles 10/50/90/99: {np.percentile(dur,[10,50,90,99]).round(2)}")
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
names = ["loop top", "load+range", "b1 wait", "image+hist", "b2 wait", "prefixA", "b3+prefixB+b4", "scatter", "b5 wait", "members+fold"]
for w, nm in ((0, "oldest wave"), (1, "youngest wave")):
    st = d2[:, w, 4:14].astype(np.float64)
    print('rows through the tie path per workgroup (mean):', (st[:, 0] // 1000000).mean())
    st[:, 0] = st[:, 0] % 1000000
    two = rows == 2
    print(f"phase cycles ({nm}; median over workgroups with 2 rows, both rows together):")
    print("   " + " | ".join(f"{n_} {np.median(st[two, i]):.0f}" for i, n_ in enumerate(names)))
