#!/usr/bin/env python3
"""One secondary workload, a few launches (for rocprofv3 / tools/pmc_rank.py).
usage: time_secondary.py simplex8|simplex3|l1|strict|band [reps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from statdepth_amd import engine, _native
if os.environ.get("SD_LIB"):                     # experiments: another build of the library (this tool only)
    _native.LIB_PATH = os.path.abspath(os.environ["SD_LIB"])
what = sys.argv[1] if len(sys.argv) > 1 else "simplex8"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.default_rng(1236)
if what == "simplex8":
    P = torch.from_numpy(rng.normal(size=(500, 50, 8)).cumsum(axis=1)).cuda()
    fn = lambda: engine.multi_simplex_counts(P, None, relax=True, samples=256, seed=1236)
    units, label = 500 * 256 * 50, "simplex tests (9 x 9 systems)"
elif what == "simplex3":
    P = torch.from_numpy(rng.normal(size=(100000, 3))).cuda()
    fn = lambda: engine.pointcloud_simplex_counts(P, samples=256, seed=1237)
    units, label = 100000 * 256, "simplex tests (4 x 4 systems)"
elif what == "l1":
    P = torch.from_numpy(rng.normal(size=(100000, 3))).cuda()
    fn = lambda: engine.l1_depth(P)
    units, label = 1e10, "point pairs"
elif what == "strict":
    X = torch.from_numpy(np.sort(rng.normal(size=2000))[None, :] * 3.0 + rng.normal(size=(1000, 2000)) * 0.3).cuda()
    fn = lambda: engine.bd_strict_counts(X)
    units, label = 2000 * 1999 * 1998 / 2, "pair tests over all timepoints"
else:
    P = torch.from_numpy(rng.normal(size=(5000, 500, 8)).cumsum(axis=1)).cuda()
    fn = lambda: engine.multi_band_counts(P)
    units, label = 5000.0 * 4999 * 4998 / 2 * 500, "pair-timepoint containment tests (equivalent)"
fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"{what}: {ms:.3f} ms per call (incl. the D2H of the result), {units / (ms * 1e-3):.3e} {label}/s", flush=True)
